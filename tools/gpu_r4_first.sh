#!/bin/bash
# round-4 first call: GPU tests, the two-rank rehearsal of the N > 1 bench step (gloo plane, and the RCCL plane's fallback), the default bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc $?" | tee -a $O/gpu_tests.log
tail -n 3 $O/gpu_tests.log
PSVR_BENCH_REHEARSE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --pairs 250000 --cpu-pairs 0 --no-e2e --no-cfg5 > $O/rehearse_gloo.log 2>&1; echo "rehearse rc $?"
tail -n 1 $O/rehearse_gloo.log | cut -c1-1500
PSVR_BENCH_REHEARSE=1 PSVR_BENCH_REHEARSE_TRY_RCCL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --pairs 100000 --cpu-pairs 0 --no-e2e --no-cfg5 > $O/rehearse_rccl_fallback.log 2>&1; echo "rehearse-rccl rc $?"
tail -n 1 $O/rehearse_rccl_fallback.log | cut -c1-800
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
cut -c1-600 $O/bench.json
