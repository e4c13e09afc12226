#!/bin/bash
# round 2, GPU session 34: summation parts re-measured with the eight-row loop (longer workgroups, longer launch tails).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/summation_parts_ab.py 1048576 10 3 > gpurun_out/r02_s34_parts.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s34_parts.txt; echo "rc=$rc"; exit $rc
