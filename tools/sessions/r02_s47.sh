#!/bin/bash
# round 2, GPU session 47: BASELINE config 5's size on one GPU with the bodies along the Morton curve: N = 2^22, 100 steps.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/run_sharded.py --bodies 4194304 --steps 100 --energy-every 50 --softening 1e-2 --body-order morton > gpurun_out/r02_s47_longrun_4m_morton.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s47_longrun_4m_morton.txt; echo "rc=$rc"; exit $rc
