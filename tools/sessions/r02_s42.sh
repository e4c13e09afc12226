#!/bin/bash
# round 2, GPU session 42: the body order moved into the library for the multi object (nbody_multi_config.body_order):
# body-order, multi-GPU and C++ host tests.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s42_build.log 2>&1 || { tail -20 gpurun_out/r02_s42_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_body_order.py tests/test_multi_gpu.py tests/test_host_cli_gpu.py -m gpu -x -q -k "not config5" > gpurun_out/r02_s42_tests.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_s42_tests.txt; echo "rc=$rc"; exit $rc
