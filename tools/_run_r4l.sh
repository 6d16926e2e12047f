cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4l && timeout -k 10 700 python -m pytest tests/test_aln_gpu.py tests/test_abi_gpu.py tests/test_shard_gpu.py -x -q -m gpu > gpurun_out/r4l/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -n 3 gpurun_out/r4l/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu_ab_env.sh r4l 2 "rows=" "rows_view=PSVR_VCNT_VIEW=1"
cd /tmp && export TMPDIR=/tmp
B="--cpu-pairs 0 --check-pairs 0 --no-e2e --no-cfg5 --one-pass"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4l/ks -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $B > $GRAFT_REPO_ROOT/gpurun_out/r4l/ks.log 2>&1; echo "trace rc $?"
