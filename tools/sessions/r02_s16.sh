#!/bin/bash
# round 2, GPU session 16: sustained runs of the final code on one GPU through the library-owned multi object (one RCCL rank):
# N = 2^20, 1000 steps, energy every 100; N = 2^22 (BASELINE config 5's size), 100 steps, energy every 50.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s16_build.log 2>&1 || { tail -20 gpurun_out/r02_s16_build.log; exit 1; }
timeout -k 10 500 python tools/run_sharded.py --bodies 1048576 --steps 1000 --energy-every 100 --softening 1e-3 > gpurun_out/r02_s16_longrun_1m.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s16_longrun_1m.txt; echo "rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python tools/run_sharded.py --bodies 4194304 --steps 100 --energy-every 50 --softening 1e-2 > gpurun_out/r02_s16_longrun_4m.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s16_longrun_4m.txt; echo "rc=$rc"
