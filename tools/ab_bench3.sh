#!/bin/bash
# GPU box: three interleaved rounds of the quick bench (no e2e / cfg5 / CPU legs) for every build/exp/lib_*.so variant: ms per step each
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
cp $R/pansvr_amd/libpsvr_engine.so $R/build/exp/_orig.so
for round in 1 2 3; do
  for f in $R/build/exp/lib_*.so; do
    n=$(basename $f .so)
    cp $f $R/pansvr_amd/libpsvr_engine.so
    timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-e2e --no-cfg5 --cpu-pairs 0 --check-pairs 20000 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$n round $round ms/step', j['ms_per_step'], 'walk_us', j['engine'].get('walk_us'), 'differing', j['parity_check']['pairs_differing'])"
  done
done
cp $R/build/exp/_orig.so $R/pansvr_amd/libpsvr_engine.so
