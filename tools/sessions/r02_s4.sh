#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s4_build.log 2>&1 || { tail -20 gpurun_out/r02_s4_build.log; exit 1; }
timeout -k 10 300 python tools/stream_overlap_probe.py 8 > gpurun_out/r02_s4_overlap8.txt 2>&1; rc=$?; cat gpurun_out/r02_s4_overlap8.txt; [ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/stream_overlap_probe.py 16 > gpurun_out/r02_s4_overlap16.txt 2>&1; rc=$?; cat gpurun_out/r02_s4_overlap16.txt; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "per_particle or equal_mass" > gpurun_out/r02_s4_pytest.log 2>&1; tail -3 gpurun_out/r02_s4_pytest.log
