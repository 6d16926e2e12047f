#!/bin/bash
# builds variants of the WHOLE libpsvr_engine.so that differ in -D defines seen by all three HIP sources (planner constants such as the
# team kernel's shape live in ksw_device.h):  tools/ab_build_full.sh name "-DX=1 -DY=2" ...   -> build/exp/lib_<name>.so (tools/ab_bench.sh)
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/exp
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  objs=""
  for s in engine ksw_host ksw_kernels; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $defs -c $R/pansvr_amd/csrc/$s.hip -o $R/build/exp/${s}_$name.o &
    objs="$objs $R/build/exp/${s}_$name.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/exp/lib_$name.so $objs
  echo built $name "($defs)"
done
