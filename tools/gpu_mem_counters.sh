set -e
# L1/L2 traffic counters per kernel for one bench step (separate --pmc passes)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/memc
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/memc/p$i -o p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/memc/p$i.log 2>&1 || { echo "set $i failed"; tail -n 5 $R/gpurun_out/memc/p$i.log; }
done
ls $R/gpurun_out/memc
