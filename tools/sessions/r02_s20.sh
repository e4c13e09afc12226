#!/bin/bash
# experiment: the packed two-columns-per-step loop of the equal-mass pair-once tiles (NBODY_SYM_PACKED=1)
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s20_build.log 2>&1 || { tail -20 gpurun_out/r02_s20_build.log; exit 1; }
NBODY_SYM_PACKED=1 timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "equal_mass or symmetric or pair_once or headline or smoke or graph_replay or config2" > gpurun_out/r02_s20_pytest.log 2>&1
rc=$?; tail -8 gpurun_out/r02_s20_pytest.log; echo "pytest(packed) rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
NBODY_SYM_PACKED=0 timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 3 --split-len 1024 --libs scalar=n_body_problem_amd/libnbody_amd.so 2>&1 | tail -1
NBODY_SYM_PACKED=1 timeout -k 10 200 python tools/ab_force.py --symmetric --rpl 0 --rounds 3 --split-len 1024 --libs packed=n_body_problem_amd/libnbody_amd.so 2>&1 | tail -1
done > gpurun_out/r02_s20_ab_packed.txt 2>&1
cat gpurun_out/r02_s20_ab_packed.txt
