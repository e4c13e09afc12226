#!/bin/bash
# round 2, GPU session 37: does the ORDER of the bodies change the force pass time (operand toggling -> power -> clock)?
# The same Plummer sphere as generated (random order), along a Morton curve, by radius.
set -o pipefail
mkdir -p gpurun_out
{ timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 --order given &&
  timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 --order morton &&
  timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 --order radius &&
  timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 5 --split-len 1024 --order given ; } > gpurun_out/r02_s37_order.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s37_order.txt; echo "rc=$rc"; exit $rc
