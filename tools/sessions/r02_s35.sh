#!/bin/bash
# round 2, GPU session 35: split length 1024 against 2048 with the eight-row loop, interleaved and sustained (N = 2^20,
# 20 steps x 5 rounds), and at N = 786432.
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 400 python tools/split_len_ab.py 1048576 20 5 1024 2048 &&
  timeout -k 10 300 python tools/split_len_ab.py 786432 20 3 1024 2048 ; } > gpurun_out/r02_s35_split_len.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s35_split_len.txt; echo "rc=$rc"; exit $rc
