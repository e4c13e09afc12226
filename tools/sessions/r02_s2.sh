#!/bin/bash
# round 2, GPU session 2: equal-mass inner loop -- tests, then the A/B at N = 2^20.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s2_build.log 2>&1 || { tail -20 gpurun_out/r02_s2_build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r02_s2_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r02_s2_pytest.log; echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/ab_equal_mass.py 3 > gpurun_out/r02_s2_ab_equal_mass.txt 2>&1
rc=$?; cat gpurun_out/r02_s2_ab_equal_mass.txt | head -12; echo "ab rc=$rc"
