#!/bin/bash
# round 2, GPU session 52: experiment: the odd summation parts' tiles on a second, lowest-priority stream (their head fills
# the tail of the part before).  Bit identity through the summation-parts tests, then step times with and without.
set -o pipefail
mkdir -p gpurun_out
NBODY_SYM_TILE_STREAMS=2 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "summation_parts or headline or full_size" > gpurun_out/r02_s52_tests.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s52_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
for round in 1 2 3; do
  NBODY_SYM_TILE_STREAMS=1 timeout -k 10 200 python tools/split_len_ab.py 1048576 20 1 2048 || exit 1
  NBODY_SYM_TILE_STREAMS=2 timeout -k 10 200 python tools/split_len_ab.py 1048576 20 1 2048 || exit 1
done > gpurun_out/r02_s52_two_streams.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s52_two_streams.txt; echo "rc=$rc"; exit $rc
