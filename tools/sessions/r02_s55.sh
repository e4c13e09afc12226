#!/bin/bash
# round 2, GPU session 55: the lane-to-row assignment in the one-sided packed kernel: same bits as the build before, the
# force pass against it (bodies along the Morton curve), and the one-sided parity tests.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_bits.py new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so > gpurun_out/r02_s55_bits.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s55_bits.txt; echo "bits rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/ab_force.py --rpl 0 --rounds 5 --order morton --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so > gpurun_out/r02_s55_ab.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s55_ab.txt; echo "ab rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > gpurun_out/r02_s55_tests.txt 2>&1
rc=$?; tail -3 gpurun_out/r02_s55_tests.txt; echo "rc=$rc"; exit $rc
