#!/bin/bash
# round 2, GPU session 50: automatic number of summation parts: the parity suite, a short randomized cross-check, and the
# step time at N = 524288 and 262144 (one launch now) against 8 parts.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_body_order.py -m gpu -x -q > gpurun_out/r02_s50_tests.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s50_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/fuzz_gpu.py 20 1729 > gpurun_out/r02_s50_fuzz.txt 2>&1
rc=$?; tail -2 gpurun_out/r02_s50_fuzz.txt; echo "fuzz rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/summation_parts_ab.py 524288 20 2 > gpurun_out/r02_s50_parts_512k.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s50_parts_512k.txt; echo "rc=$rc"; exit $rc
