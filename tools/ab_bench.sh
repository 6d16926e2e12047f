#!/bin/bash
# GPU box: bench every build/exp/lib_*.so variant (see tools/ab_build.sh) on the 1 M-pair workload, one line each
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
cp $R/pansvr_amd/libpsvr_engine.so $R/build/exp/_orig.so
for f in $R/build/exp/lib_*.so; do
  n=$(basename $f .so)
  cp $f $R/pansvr_amd/libpsvr_engine.so
  timeout -k 10 240 python3 $R/bench.py --steps 5 --cpu-pairs 20000 --check-pairs 20000 > $R/gpurun_out/ab/$n.json 2> $R/gpurun_out/ab/$n.err || { echo "$n FAILED"; cp $R/build/exp/_orig.so $R/pansvr_amd/libpsvr_engine.so; exit 1; }
  python3 - $R/gpurun_out/ab/$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels_ms_per_step", {})
print(sys.argv[2], "ms/step", d["ms_per_step"], "team", k.get("extd2_team_kernel"), "tiny", k.get("extd2_tiny_kernel"), "differing", d.get("parity_check", {}).get("pairs_differing"))
PY
done
cp $R/build/exp/_orig.so $R/pansvr_amd/libpsvr_engine.so
