#!/bin/bash
# round 2, GPU session 44: refreshing the Morton layout (nbody_multi_reorder, the period, the Python layer's reorder): the
# tests, then 1000 steps at N = 2^20 with a refresh every 100 steps (against the run without, session 43).
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s44_build.log 2>&1 || { tail -20 gpurun_out/r02_s44_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_body_order.py tests/test_multi_gpu.py tests/test_host_cli_gpu.py -m gpu -x -q -k "not config5" > gpurun_out/r02_s44_tests.txt 2>&1
rc=$?; tail -8 gpurun_out/r02_s44_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/run_sharded.py --bodies 1048576 --steps 1000 --energy-every 100 --softening 1e-3 --body-order morton --reorder-every 100 > gpurun_out/r02_s44_longrun_reorder100.txt 2>&1
rc=$?; tail -12 gpurun_out/r02_s44_longrun_reorder100.txt; echo "rc=$rc"; exit $rc
