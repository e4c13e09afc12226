#!/bin/bash
# round 2, GPU session 29: what the LDS traffic of the packed pair-once loop costs (timing-only variants with wrong sums:
# no permutes, no column reads, neither) -- decides whether 8 rows per lane (half the LDS operations per pair) can pay.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 \
  --libs base=n_body_problem_amd/libnbody_amd.so,norot=build/variants/libnbody_exp_norot.so,noread=build/variants/libnbody_exp_noread.so,nolds=build/variants/libnbody_exp_nolds.so > gpurun_out/r02_s29_ab.txt 2>&1
rc=$?; cat gpurun_out/r02_s29_ab.txt; echo "ab rc=$rc"; exit $rc
