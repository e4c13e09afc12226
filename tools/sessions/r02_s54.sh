#!/bin/bash
# round 2, GPU session 54: the lane-to-body assignment in every hand-scheduled pair-once loop: the whole GPU suite and two
# seeds of the randomized cross-check.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_s54_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r02_s54_pytest.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py 40 2718 > gpurun_out/r02_s54_fuzz_a.txt 2>&1
rc=$?; tail -2 gpurun_out/r02_s54_fuzz_a.txt; echo "fuzz rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py 40 1618 > gpurun_out/r02_s54_fuzz_b.txt 2>&1
rc=$?; tail -2 gpurun_out/r02_s54_fuzz_b.txt; echo "fuzz rc=$rc"; exit $rc
