#!/bin/bash
# GPU box: interleaved rounds of the quick bench (no e2e / cfg5 / CPU legs / pipeline) under different environments.
# usage: bash tools/gpu_ab_env.sh <tag> <rounds> "NAME=ENV1=1 ENV2=2" "NAME2=" ...     (a variant is  name=<space separated env assignments>)
R=$GRAFT_REPO_ROOT
TAG=$1; ROUNDS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p $O
for round in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}
    env $envs timeout -k 10 240 python3 $R/bench.py --steps 10 --warmup 3 --no-e2e --no-cfg5 --no-pipeline --cpu-pairs 0 --check-pairs 20000 2>$O/$name.$round.err | tail -n 1 > $O/$name.$round.json
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
