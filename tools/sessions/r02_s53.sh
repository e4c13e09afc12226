#!/bin/bash
# round 2, GPU session 53: the eight-row loop with consecutive bodies in the four lanes one ALU lane serves (operand
# toggling again): parity / sharding / body-order tests, then the force pass against the previous build, bodies along the
# Morton curve and in the generator's order.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py tests/test_body_order.py -m gpu -x -q > gpurun_out/r02_s53_tests.txt 2>&1
rc=$?; tail -4 gpurun_out/r02_s53_tests.txt; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
{ timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0 --rounds 6 --split-len 2048 --order morton --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so &&
  timeout -k 10 300 python tools/ab_force.py --symmetric --rpl 0 --rounds 4 --split-len 2048 --order given --libs new=n_body_problem_amd/libnbody_amd.so,prev=build/variants/libnbody_prev.so ; } > gpurun_out/r02_s53_ab.txt 2>&1
rc=$?; grep -v amdgpu.ids gpurun_out/r02_s53_ab.txt; echo "ab rc=$rc"; exit $rc
