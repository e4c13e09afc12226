#!/bin/bash
# round 2, GPU session 31: the eight-row kernel where it is not at home: general masses at N = 2^20 (its non-uniform tiles
# take the four-row loop with two waves per workgroup), and short splits (N = 131072 / 512, N = 409600 / 2048).
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0,8 --rounds 4 --split-len 1024 --general-masses &&
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0,8 --rounds 8 --n 131072 --split-len 512 &&
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0,8 --rounds 8 --n 262144 --split-len 1024 &&
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0,8 --rounds 4 --n 1048576 --split-len 2048 ; } > gpurun_out/r02_s31_ab.txt 2>&1
rc=$?; cat gpurun_out/r02_s31_ab.txt; echo "ab rc=$rc"; exit $rc
