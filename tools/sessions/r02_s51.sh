#!/bin/bash
# round 2, GPU session 51: the final configuration sustained: N = 2^20 (2048-body splits, 4 summation parts), 1000 steps,
# bodies along the Morton curve refreshed every 100 steps.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/run_sharded.py --bodies 1048576 --steps 1000 --energy-every 100 --softening 1e-3 --body-order morton --reorder-every 100 > gpurun_out/r02_s51_longrun_final.txt 2>&1
rc=$?; tail -12 gpurun_out/r02_s51_longrun_final.txt; echo "rc=$rc"; exit $rc
