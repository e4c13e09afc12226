#!/bin/bash
# round 2, GPU session 6: randomized cross-check (mass patterns, library-owned exchange) and the gap / priority variants of
# the pair-once tile loop on the equal-mass path.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s6_build.log 2>&1 || { tail -20 gpurun_out/r02_s6_build.log; exit 1; }
timeout -k 10 900 python tools/fuzz_gpu.py 60 2027 > gpurun_out/r02_s6_fuzz.txt 2>&1
rc=$?; tail -6 gpurun_out/r02_s6_fuzz.txt; echo "fuzz rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/ab_force.py --symmetric --rpl 0 --rounds 4 --split-len 1024 \
  --libs base=n_body_problem_amd/libnbody_amd.so,nogap=build/variants/libnbody_nogap.so,gap11=build/variants/libnbody_gap11.so,noprio=build/variants/libnbody_noprio.so \
  > gpurun_out/r02_s6_ab_variants.txt 2>&1
rc=$?; cat gpurun_out/r02_s6_ab_variants.txt | tail -8; echo "ab rc=$rc"
