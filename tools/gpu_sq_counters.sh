set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq/a -o a -- python3 $R/tools/dp_bench.py 200000 35 94 > $R/gpurun_out/sq/a.log 2>&1
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq/b -o b -- python3 $R/tools/dp_bench.py 200000 35 94 > $R/gpurun_out/sq/b.log 2>&1
tail -n 3 $R/gpurun_out/sq/a.log; tail -n 3 $R/gpurun_out/sq/b.log
