#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s21_build.log 2>&1 || { tail -20 gpurun_out/r02_s21_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -k "not config5" > gpurun_out/r02_s21_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r02_s21_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python tools/fuzz_gpu.py 40 777 > gpurun_out/r02_s21_fuzz.txt 2>&1; rc=$?; tail -2 gpurun_out/r02_s21_fuzz.txt; echo "fuzz rc=$rc"
timeout -k 10 600 python tools/ab_force.py --symmetric --rpl 0 --rounds 4 --split-len 1024 \
  --libs base=n_body_problem_amd/libnbody_amd.so,nogap=build/variants/libnbody_pknogap.so,gap11=build/variants/libnbody_pkgap11.so,noprio=build/variants/libnbody_pknoprio.so \
  > gpurun_out/r02_s21_ab_variants.txt 2>&1
cat gpurun_out/r02_s21_ab_variants.txt | tail -5
