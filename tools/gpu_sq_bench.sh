set -e
# SQ instruction / wait counters of every kernel of one bench step (two passes; counters only with --kernel-trace)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sqb
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sqb/a -o a -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/sqb/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sqb/b -o b -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/sqb/b.log 2>&1
tail -n 2 $R/gpurun_out/sqb/a.log; tail -n 2 $R/gpurun_out/sqb/b.log
