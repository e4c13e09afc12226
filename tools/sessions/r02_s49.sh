#!/bin/bash
# round 2, GPU session 49: parity, sharded and multi suites after nbody_invalidate_forces learned to drop stale partial sums.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_sharded_gpu.py tests/test_multi_gpu.py tests/test_body_order.py -m gpu -x -q > gpurun_out/r02_s49_tests.txt 2>&1
rc=$?; tail -6 gpurun_out/r02_s49_tests.txt; echo "rc=$rc"; exit $rc
