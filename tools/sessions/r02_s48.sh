#!/bin/bash
# round 2, GPU session 48: the randomized cross-check with the body order as one more random choice of the multi-GPU leg.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/fuzz_gpu.py 50 31415 > gpurun_out/r02_s48_fuzz.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s48_fuzz.txt; echo "fuzz rc=$rc"; exit $rc
