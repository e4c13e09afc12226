#!/bin/bash
# round 2, GPU session 26: the column sums exchanged as own-row segments (not gathered whole): multi-GPU suite and the
# randomized cross-check; then one rank of 8's share of a step at N = 2^20 (both stream layouts) and of N = 2^22.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s26_build.log 2>&1 || { tail -20 gpurun_out/r02_s26_build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_multi_gpu.py tests/test_sharded_gpu.py tests/test_host_cli_gpu.py -m gpu -x -q > gpurun_out/r02_s26_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_s26_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py 25 5150 > gpurun_out/r02_s26_fuzz.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_s26_fuzz.txt; echo "fuzz rc=$rc"; [ $rc -ne 0 ] && exit $rc
{ timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 1024 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 1024 &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 2 --rank 1 --split-len 1024 --two-streams &&
  timeout -k 10 300 python tools/shard_rate.py --bodies 4194304 --world 8 --rank 3 --split-len 2048 1024 --two-streams ; } > gpurun_out/r02_s26_shard_rate.txt 2>&1
rc=$?; cat gpurun_out/r02_s26_shard_rate.txt; echo "rc=$rc"; exit $rc
