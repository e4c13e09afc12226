#!/bin/bash
# round 2, GPU session 30: the eight-rows-per-lane loop for equal-mass pair-once tiles (NBODY_SYM_PACKED=2): parity and
# sharding tests with it, then the force pass against the four-row loop in one process (rows per lane 0 = default, 8).
set -o pipefail
mkdir -p gpurun_out
NBODY_SYM_PACKED=2 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q > gpurun_out/r02_s30_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_s30_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/ab_force.py --symmetric --rpl 0,8 --rounds 6 --split-len 1024 > gpurun_out/r02_s30_ab.txt 2>&1
rc=$?; cat gpurun_out/r02_s30_ab.txt; echo "ab rc=$rc"; exit $rc
