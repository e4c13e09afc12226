#!/bin/bash
# round 2, GPU session 46: bench.py's own rank launcher on the final tree: two ranks sharing cuda:0 over gloo (the rehearsal
# harness), and --gpus 2 on this one-GPU box with the default backend (must refuse, non-zero exit).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --steps 3 --warmup 1 --n 262144 --no-cpu-baseline > gpurun_out/r02_s46_bench_gloo2.json 2> gpurun_out/r02_s46_bench_gloo2.err
rc=$?; tail -c 500 gpurun_out/r02_s46_bench_gloo2.json; echo; tail -3 gpurun_out/r02_s46_bench_gloo2.err; echo "gloo2 rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02_s46_bench_refuse.json 2> gpurun_out/r02_s46_bench_refuse.err
rc=$?; tail -3 gpurun_out/r02_s46_bench_refuse.err; echo "refuse rc=$rc (expected non-zero)"; [ $rc -eq 0 ] && exit 1
exit 0
