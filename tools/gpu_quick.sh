#!/bin/bash
# a quick look at a change: the engine's parity tests, the bench line without its side legs (twice), a kernel trace of a few steps
R=$GRAFT_REPO_ROOT
TAG=${1:-quick}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_aln_gpu.py tests/test_abi_gpu.py -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc $rc"
tail -n 2 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for k in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-pairs 0 --no-e2e --no-cfg5 > $O/bench$k.json 2> $O/bench$k.err; rc=$?; echo "bench rc $rc"
python - $O/bench$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k:v for k,v in d["engine"].items() if "walk" in k})
PY
[ $rc -eq 0 ] || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-pairs 0 --no-e2e --no-cfg5 --one-pass > $O/ks.log 2>&1; echo "trace rc $?"
find $O/ks -name '*kernel_trace.csv' | head -n 1
