#!/bin/bash
# round-4 baseline of the restored tree: the GPU tests, the default bench line, the two-rank rehearsal, kernel stats + PMC passes
R=$GRAFT_REPO_ROOT
TAG=${1:-r4e}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc $rc" | tee -a $O/gpu_tests.log
tail -n 3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc $rc"
cut -c1-400 $O/bench.json
[ $rc -eq 0 ] || exit 1
PSVR_BENCH_REHEARSE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --pairs 250000 --cpu-pairs 0 --no-e2e --no-cfg5 > $O/rehearse_gloo.log 2>&1; echo "rehearse rc $?"
tail -n 1 $O/rehearse_gloo.log | cut -c1-600
bash tools/gpu_round_profile.sh $TAG > $O/profile.log 2>&1; echo "profile rc $?"
tail -n 5 $O/profile.log
