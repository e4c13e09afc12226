#!/bin/bash
# round 2, GPU session 56: nbody_run --morton --reorder-every on one device against the Python layer's schedule.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s56_build.log 2>&1 || { tail -20 gpurun_out/r02_s56_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_host_cli_gpu.py -m gpu -x -q -k "not config5" > gpurun_out/r02_s56_tests.txt 2>&1
rc=$?; tail -6 gpurun_out/r02_s56_tests.txt; echo "rc=$rc"; exit $rc
