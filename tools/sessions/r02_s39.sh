#!/bin/bash
# round 2, GPU session 39: Hilbert against Morton order (force pass time of the same sphere), alternating.
set -o pipefail
mkdir -p gpurun_out
{ for o in morton hilbert morton hilbert given; do timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 6 --split-len 1024 --order $o || exit 1; done ; } > gpurun_out/r02_s39_order.txt 2>&1
rc=$?; grep -v "amdgpu.ids\|^N=" gpurun_out/r02_s39_order.txt; echo "rc=$rc"; exit $rc
