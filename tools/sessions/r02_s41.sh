#!/bin/bash
# round 2, GPU session 41: the multi-GPU suite after the torch-nccl test grew a Morton-stored leg.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_multi_gpu.py -m gpu -x -q > gpurun_out/r02_s41_tests.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_s41_tests.txt; echo "rc=$rc"; exit $rc
