#!/bin/bash
# Build the library as of a git revision (default HEAD) for A/B runs: tools/build_prev.sh [rev] -> build/variants/libnbody_prev.so
set -e
rev=${1:-HEAD}
cd "$(dirname "$0")/.."
tmp=build/variants/src_prev
rm -rf $tmp; mkdir -p $tmp/n_body_problem_amd/csrc $tmp/include build/variants/obj_prev
for f in nbody_kernels.hip nbody_symmetric.hip nbody_capi.hip nbody_multi.hip nbody_kernels.h; do git show $rev:n_body_problem_amd/csrc/$f > $tmp/n_body_problem_amd/csrc/$f; done
git show $rev:include/nbody.h > $tmp/include/nbody.h
for f in nbody_kernels nbody_symmetric nbody_capi nbody_multi; do
  extra=""
  [ $f = nbody_kernels -o $f = nbody_symmetric ] && extra="-fno-slp-vectorize"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $extra -c $tmp/n_body_problem_amd/csrc/$f.hip -o build/variants/obj_prev/$f.o
done
hipcc -shared -fPIC --offload-arch=gfx950 build/variants/obj_prev/*.o -L/opt/rocm/lib -lrccl -o build/variants/libnbody_prev.so
echo build/variants/libnbody_prev.so
