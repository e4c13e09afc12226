#!/bin/bash
# round 2, GPU session 28: packed pair-once loops unrolled by four steps with immediate LDS offsets (x[128] staging) against
# the previous build: same bits, then force pass times interleaved (equal-mass and general-mass tiles).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_bits.py new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so > gpurun_out/r02_s28_bits.txt 2>&1
rc=$?; cat gpurun_out/r02_s28_bits.txt; echo "bits rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/ab_force.py --symmetric --rpl 0 --rounds 6 --split-len 1024 \
  --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so > gpurun_out/r02_s28_ab.txt 2>&1
rc=$?; cat gpurun_out/r02_s28_ab.txt; echo "ab rc=$rc"; [ $rc -ne 0 ] && exit $rc
NBODY_AB_GENERAL=1 timeout -k 10 500 python tools/ab_force.py --symmetric --rpl 0 --rounds 4 --split-len 1024 --general-masses \
  --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so > gpurun_out/r02_s28_ab_general.txt 2>&1
rc=$?; cat gpurun_out/r02_s28_ab_general.txt; echo "ab general rc=$rc"; exit $rc
