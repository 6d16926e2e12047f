#!/bin/bash
# variants of the whole libpsvr_engine.so that differ in -D defines, into gpu_exp/ (travels to the GPU box; git-ignored):
#   tools/ab_build_exp.sh name "-DX=1" name2 "-DY=2" ...   then on the box: bash tools/gpu_ab_libs.sh <tag> <rounds> [ENV=1 ...]
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpu_exp $R/build/exp
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  objs=""
  for s in engine ksw_host ksw_kernels bgzf; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $defs -c $R/pansvr_amd/csrc/$s.hip -o $R/build/exp/${s}_$name.o &
    objs="$objs $R/build/exp/${s}_$name.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpu_exp/lib_$name.so $objs
  echo built $name "($defs)"
done
