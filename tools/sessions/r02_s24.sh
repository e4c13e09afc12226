#!/bin/bash
# round 2, GPU session 24: summation parts (1/2/4/8 launches of whole row groups, two buffer slots from 4 on): the two new
# tests, the multi-GPU suite (shards take one part), then time and memory per number of parts at N = 2^20.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s24_build.log 2>&1 || { tail -20 gpurun_out/r02_s24_build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q > gpurun_out/r02_s24_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_s24_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/summation_parts_ab.py 1048576 10 3 > gpurun_out/r02_s24_parts.txt 2>&1
rc=$?; cat gpurun_out/r02_s24_parts.txt; echo "rc=$rc"; exit $rc
