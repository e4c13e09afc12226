#!/bin/bash
# round 2, GPU session 38: bodies stored along a Morton curve (body_order): the new tests, the C++ host's --morton, then
# the bench with both orders in one run.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s38_build.log 2>&1 || { tail -20 gpurun_out/r02_s38_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_body_order.py tests/test_host_cli_gpu.py -m gpu -x -q -k "not config5" > gpurun_out/r02_s38_tests.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_s38_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r02_s38_bench.json 2> gpurun_out/r02_s38_bench.err
rc=$?; tail -c 300 gpurun_out/r02_s38_bench.json; echo "bench rc=$rc"; exit $rc
