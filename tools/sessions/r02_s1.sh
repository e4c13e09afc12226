#!/bin/bash
# round 2, GPU session 1: full GPU suite, smoke, bench (1 GPU), bench self-launch rehearsal (2 ranks on one GPU over gloo),
# and what RCCL says to two ranks on one device.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s1_build.log 2>&1 || { tail -20 gpurun_out/r02_s1_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=8 -x --durations=15 > gpurun_out/r02_s1_pytest.log 2>&1
rc=$?; tail -25 gpurun_out/r02_s1_pytest.log; echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r02_s1_smoke.log 2>&1
rc=$?; tail -3 gpurun_out/r02_s1_smoke.log; echo "smoke rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r02_s1_bench.json 2> gpurun_out/r02_s1_bench.err
rc=$?; tail -c 1500 gpurun_out/r02_s1_bench.json; echo "bench rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --gpus 2 --single-device --backend gloo --steps 2 --no-cpu-baseline > gpurun_out/r02_s1_bench_2rank.json 2> gpurun_out/r02_s1_bench_2rank.err
rc=$?; tail -c 600 gpurun_out/r02_s1_bench_2rank.json; tail -5 gpurun_out/r02_s1_bench_2rank.err; echo "bench 2-rank rehearsal rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
python bench.py --gpus 2 --steps 1 > gpurun_out/r02_s1_bench_2gpu_refused.json 2> gpurun_out/r02_s1_bench_2gpu_refused.err; echo "bench --gpus 2 on a 1-GPU box: rc=$? (must be non-zero)"; tail -2 gpurun_out/r02_s1_bench_2gpu_refused.err
timeout -k 10 180 python tools/rccl_two_ranks_one_device.py > gpurun_out/r02_s1_rccl_dup.log 2>&1
echo "rccl dup-device probe rc=$?"; tail -8 gpurun_out/r02_s1_rccl_dup.log
