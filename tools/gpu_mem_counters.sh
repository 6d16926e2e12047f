set -e
# L1/L2 traffic counters per kernel for one bench step (separate --pmc passes, ~2.5 min each: every kernel of the index build is counted too; the same
# command line as tools/gpu_round_profile.sh -- with a warm-up step or the parity sample a pass does not fit 300 s); a per-kernel table at the end
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-memc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 --check-pairs 0 --no-e2e --no-cfg5 --one-pass > $O/p$i.log 2>&1 || { echo "set $i failed"; tail -n 5 $O/p$i.log; }
done
python3 - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
tab=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(O+'/p*/**/*counter_collection.csv',recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('psvr::','')[:34]
        tab[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[k]+=1
names=sorted({c for k in tab for c in tab[k]})
print('kernel',*names,sep=' | ')
for k in sorted(tab,key=lambda k:-tab[k].get('GRBM_GUI_ACTIVE',0))[:16]:
    print(k,*['%.4g'%tab[k].get(c,0) for c in names],sep=' | ')
PY
