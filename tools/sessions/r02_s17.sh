#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s17_build.log 2>&1 || { tail -20 gpurun_out/r02_s17_build.log; exit 1; }
for L in 1024 2048 4096; do
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0 --rounds 3 --split-len $L --libs base=n_body_problem_amd/libnbody_amd.so > gpurun_out/r02_s17_split_$L.txt 2>&1
  tail -1 gpurun_out/r02_s17_split_$L.txt
done
timeout -k 10 900 python tools/fuzz_gpu.py 60 424242 > gpurun_out/r02_s17_fuzz.txt 2>&1; rc=$?; tail -2 gpurun_out/r02_s17_fuzz.txt; echo "fuzz rc=$rc"
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_s17_pytest.log 2>&1; tail -3 gpurun_out/r02_s17_pytest.log
