#!/bin/bash
# builds variants of libpsvr_engine.so that differ in -D defines of ksw_kernels.hip:  tools/ab_build.sh name "-DX=1 -DY=2" ...
# (engine.o / ksw_host.o come from the last regular build); run them with tools/ab_bench.sh on the GPU box
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/exp
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $defs -c $R/pansvr_amd/csrc/ksw_kernels.hip -o $R/build/exp/ksw_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/exp/lib_$name.so $R/build/engine.o $R/build/ksw_host.o $R/build/exp/ksw_$name.o
  echo built $name "($defs)"
done
