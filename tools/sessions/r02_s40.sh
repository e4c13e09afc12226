#!/bin/bash
# round 2, GPU session 40: the body-order tests with the headline-size one.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_body_order.py -m gpu -x -q --durations=3 > gpurun_out/r02_s40_tests.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_s40_tests.txt; echo "rc=$rc"; exit $rc
