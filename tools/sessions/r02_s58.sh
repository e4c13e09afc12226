#!/bin/bash
# round 2, GPU session 58: one rank of P's share of a step with the final kernels and split lengths (the rule's: 2048).
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 1 --rank 0 --split-len 0 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 2 --rank 1 --split-len 0 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 4 --rank 2 --split-len 0 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 0 --two-streams &&
  timeout -k 10 300 python tools/shard_rate.py --bodies 4194304 --world 8 --rank 3 --split-len 0 --two-streams ; } > gpurun_out/r02_s58_shard_rate.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s58_shard_rate.txt; echo "rc=$rc"; exit $rc
