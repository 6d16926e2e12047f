set -e
# kernel trace of a short bench run (timeline / gap analysis with tools/trace_gaps.py)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace/ks -o ks -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 > $R/gpurun_out/trace/ks.log 2>&1
tail -n 2 $R/gpurun_out/trace/ks.log
