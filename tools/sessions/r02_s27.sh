#!/bin/bash
# round 2, GPU session 27: one rank of P's share of a step (force launches, summation of its groups, update) on one GPU.
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 1024 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 1024 &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 2 --rank 1 --split-len 1024 --two-streams &&
  timeout -k 10 300 python tools/shard_rate.py --bodies 4194304 --world 8 --rank 3 --split-len 2048 1024 --two-streams ; } > gpurun_out/r02_s27_shard_rate.txt 2>&1
rc=$?; cat gpurun_out/r02_s27_shard_rate.txt; echo "rc=$rc"; exit $rc
