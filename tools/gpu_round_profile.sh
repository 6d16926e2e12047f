set -e
# Evidence of one round: kernel-trace stats + per-kernel PMC counters of the bench command (separate --pmc passes, no trace domains
# besides --kernel-trace).  usage (on the GPU box): bash tools/gpu_round_profile.sh <tag>   ->  gpurun_out/<tag>/
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
B="--cpu-pairs 0 --check-pairs 0 --no-e2e --no-cfg5 --one-pass"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/bench.py --steps 2 --warmup 1 $B > $O/ks.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python3 $R/bench.py --steps 1 --warmup 0 $B > $O/p$i.log 2>&1 || { echo "pmc set $i failed"; tail -n 5 $O/p$i.log; }
done
find $O -name "*.csv" | head -20
