#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh NAME [extra hipcc flags for the kernel sources]
# -> build/variants/libnbody_NAME.so   (compare with tools/ab_force.py --libs a=...,b=...)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/variants/obj_$name
for f in nbody_kernels nbody_symmetric nbody_order nbody_capi nbody_multi; do
  extra=""
  [ $f = nbody_kernels -o $f = nbody_symmetric ] && extra="-fno-slp-vectorize"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $extra "$@" -c n_body_problem_amd/csrc/$f.hip -o build/variants/obj_$name/$f.o
done
hipcc -shared -fPIC --offload-arch=gfx950 build/variants/obj_$name/*.o -L/opt/rocm/lib -lrccl -o build/variants/libnbody_$name.so
echo build/variants/libnbody_$name.so
