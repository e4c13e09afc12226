#!/bin/bash
# round 2, GPU session 43: does the Morton layout decay?  N = 2^20, 1000 steps (one time unit), energy every 100 steps,
# bodies stored along the curve at step 0 only; then the same 300 steps in the generator's order for the box's baseline.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/run_sharded.py --bodies 1048576 --steps 1000 --energy-every 100 --softening 1e-3 --body-order morton > gpurun_out/r02_s43_longrun_morton.txt 2>&1
rc=$?; tail -12 gpurun_out/r02_s43_longrun_morton.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/run_sharded.py --bodies 1048576 --steps 300 --energy-every 100 --softening 1e-3 > gpurun_out/r02_s43_longrun_given.txt 2>&1
rc=$?; tail -5 gpurun_out/r02_s43_longrun_given.txt; echo "rc=$rc"; exit $rc
