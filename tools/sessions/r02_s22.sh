#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s22_build.log 2>&1 || { tail -20 gpurun_out/r02_s22_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -k "not config5" > gpurun_out/r02_s22_pytest.log 2>&1
rc=$?; tail -8 gpurun_out/r02_s22_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python tools/fuzz_gpu.py 60 4711 > gpurun_out/r02_s22_fuzz.txt 2>&1; rc=$?; tail -2 gpurun_out/r02_s22_fuzz.txt; echo "fuzz rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
NBODY_SYM_PACKED=1 timeout -k 10 400 python tools/ab_equal_mass.py 3 > gpurun_out/r02_s22_ab_packed1.txt 2>&1; head -7 gpurun_out/r02_s22_ab_packed1.txt
NBODY_SYM_PACKED=0 timeout -k 10 400 python tools/ab_equal_mass.py 3 > gpurun_out/r02_s22_ab_packed0.txt 2>&1; head -4 gpurun_out/r02_s22_ab_packed0.txt
