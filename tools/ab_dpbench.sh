#!/bin/bash
# GPU box: DP micro-benchmark (tools/dp_bench.py) for every build/exp/lib_*.so variant
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
cp $R/pansvr_amd/libpsvr_engine.so $R/build/exp/_orig.so
for f in $R/build/exp/lib_*.so; do
  n=$(basename $f .so)
  cp $f $R/pansvr_amd/libpsvr_engine.so
  timeout -k 10 120 python3 $R/tools/dp_bench.py "$@" > $R/gpurun_out/ab/$n.dpb 2>&1 || echo "$n FAILED"
  echo "$n $(grep 'with CIGAR' $R/gpurun_out/ab/$n.dpb | cut -c1-60) | $(grep 'score only' $R/gpurun_out/ab/$n.dpb | cut -c1-60)"
done
cp $R/build/exp/_orig.so $R/pansvr_amd/libpsvr_engine.so
