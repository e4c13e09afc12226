#!/bin/bash
# round 2, GPU session 36: the final code: smoke(), two seeds of the randomized cross-check (split lengths 1536 and 3072
# added), bench through its own rank launcher (--gpus 1 under torch.distributed.run: the RCCL leg with one rank).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > gpurun_out/r02_s36_smoke.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_s36_smoke.txt; echo "smoke rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/fuzz_gpu.py 40 8086 > gpurun_out/r02_s36_fuzz_a.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_s36_fuzz_a.txt; echo "fuzz a rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/fuzz_gpu.py 40 68000 > gpurun_out/r02_s36_fuzz_b.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_s36_fuzz_b.txt; echo "fuzz b rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 5 --warmup 1 --no-extra-legs > gpurun_out/r02_s36_bench_torchrun.json 2> gpurun_out/r02_s36_bench_torchrun.err
rc=$?; tail -c 600 gpurun_out/r02_s36_bench_torchrun.json; echo "bench torchrun rc=$rc"; exit $rc
