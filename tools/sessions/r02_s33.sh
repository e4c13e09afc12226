#!/bin/bash
# round 2, GPU session 33: idle gap and wave priorities re-measured on the eight-row loop (4 waves per SIMD).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 \
  --libs base=n_body_problem_amd/libnbody_amd.so,nogap=build/variants/libnbody_r8nogap.so,gap11=build/variants/libnbody_r8gap11.so,noprio=build/variants/libnbody_r8noprio.so > gpurun_out/r02_s33_ab.txt 2>&1
rc=$?; cat gpurun_out/r02_s33_ab.txt; echo "ab rc=$rc"; exit $rc
