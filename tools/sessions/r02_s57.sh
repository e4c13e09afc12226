#!/bin/bash
# round 2, GPU session 57: the eight-row loop's batches interleaved (rows A and B instruction by instruction): same bits as
# the build before, then the force pass against it, bodies along the Morton curve and as generated.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_bits.py new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so 1048576 > gpurun_out/r02_s57_bits.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s57_bits.txt; echo "bits rc=$rc"; [ $rc -ne 0 ] && exit $rc
{ timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0 --rounds 6 --split-len 2048 --order morton --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so &&
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0 --rounds 4 --split-len 2048 --order given --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so ; } > gpurun_out/r02_s57_ab.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s57_ab.txt; echo "ab rc=$rc"; exit $rc
