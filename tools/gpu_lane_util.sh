set -e
# VALU lane utilisation of every kernel of one bench step: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/lane
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/lane/a -o a -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/lane/a.log 2>&1
tail -n 2 $R/gpurun_out/lane/a.log
