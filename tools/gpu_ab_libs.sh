#!/bin/bash
# GPU box: interleaved rounds of the quick bench for every gpu_exp/lib_*.so variant (tools/ab_build_exp.sh); extra arguments are environment assignments
R=$GRAFT_REPO_ROOT
TAG=$1; ROUNDS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p $O
cp $R/pansvr_amd/libpsvr_engine.so $R/gpu_exp/_orig.so
for round in $(seq 1 $ROUNDS); do
  for f in $R/gpu_exp/lib_*.so; do
    name=$(basename $f .so)
    cp $f $R/pansvr_amd/libpsvr_engine.so
    env "$@" timeout -k 10 240 python3 $R/bench.py --steps 10 --warmup 3 --no-e2e --no-cfg5 --no-pipeline --cpu-pairs 0 --check-pairs 20000 2>$O/$name.$round.err | tail -n 1 > $O/$name.$round.json
    python3 - "$O/$name.$round.json" "$name" "$round" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    k=j['kernels_ms_per_step']
    print(sys.argv[2],'round',sys.argv[3],'ms/step',j['ms_per_step'],'differing',j['parity_check']['pairs_differing'],' '.join('%s=%.3f'%(a,b) for a,b in list(k.items())[:12]), flush=True)
except Exception as e:
    print(sys.argv[2],'round',sys.argv[3],'FAILED',e, flush=True)
PY
  done
done
cp $R/gpu_exp/_orig.so $R/pansvr_amd/libpsvr_engine.so
