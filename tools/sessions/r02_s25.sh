#!/bin/bash
# round 2, GPU session 25: summation parts as the default (8): randomized cross-check with several launches at every size
# (NBODY_SYM_PARTS_MIN_TILES=0 inside the tool), then the full session 5 (suite, bench, profiles of both force modes).
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s25_build.log 2>&1 || { tail -20 gpurun_out/r02_s25_build.log; exit 1; }
timeout -k 10 500 python tools/fuzz_gpu.py 30 909 > gpurun_out/r02_s25_fuzz.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s25_fuzz.txt; echo "fuzz rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/sessions/r02_s5.sh
