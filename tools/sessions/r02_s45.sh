#!/bin/bash
# round 2, GPU session 45: the whole GPU suite and smoke() on the final tree.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s45_build.log 2>&1 || { tail -20 gpurun_out/r02_s45_build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r02_s45_pytest.log 2>&1
rc=$?; tail -10 gpurun_out/r02_s45_pytest.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > gpurun_out/r02_s45_smoke.txt 2>&1
rc=$?; tail -2 gpurun_out/r02_s45_smoke.txt; echo "smoke rc=$rc"; exit $rc
