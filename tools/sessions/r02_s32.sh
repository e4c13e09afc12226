#!/bin/bash
# round 2, GPU session 32: with the eight-row loop as the default, which split length?  One GPU and one rank of 8 / of 2 at
# N = 2^20 with 1024- and 2048-body splits, N = 524288 likewise, one rank of 8 at N = 2^22 (2048 / 4096).
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 1 --rank 0 --split-len 1024 2048 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 8 --rank 3 --split-len 1024 2048 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 1048576 --world 2 --rank 1 --split-len 1024 2048 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 524288 --world 1 --rank 0 --split-len 1024 2048 --two-streams &&
  timeout -k 10 200 python tools/shard_rate.py --bodies 524288 --world 8 --rank 3 --split-len 1024 2048 --two-streams &&
  timeout -k 10 300 python tools/shard_rate.py --bodies 4194304 --world 8 --rank 3 --split-len 2048 4096 --two-streams ; } > gpurun_out/r02_s32_shard_rate.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s32_shard_rate.txt; echo "rc=$rc"; exit $rc
