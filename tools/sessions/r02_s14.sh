#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s14_build.log 2>&1 || { tail -20 gpurun_out/r02_s14_build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not config5" > gpurun_out/r02_s14_pytest.log 2>&1
rc=$?; tail -12 gpurun_out/r02_s14_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/small_n_breakdown.py > gpurun_out/r02_s14_small_n.txt 2>&1; cat gpurun_out/r02_s14_small_n.txt
timeout -k 10 300 python tools/tiny_n.py > gpurun_out/r02_s14_tiny_n.txt 2>&1; cat gpurun_out/r02_s14_tiny_n.txt
