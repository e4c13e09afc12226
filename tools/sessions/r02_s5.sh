#!/bin/bash
# round 2, GPU session 5: full GPU suite (with the config-5 dry run), bench, then the rocprofv3 profiles of both force modes.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s5_build.log 2>&1 || { tail -20 gpurun_out/r02_s5_build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r02_s5_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r02_s5_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r02_s5_bench.json 2> gpurun_out/r02_s5_bench.err
rc=$?; tail -c 400 gpurun_out/r02_s5_bench.json; echo "bench rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 bash tools/profile.sh r02_pair_once --force-mode pair_once > gpurun_out/r02_s5_profile_pair_once.log 2>&1; echo "profile pair_once rc=$?"
timeout -k 10 900 bash tools/profile.sh r02_one_sided --force-mode one_sided > gpurun_out/r02_s5_profile_one_sided.log 2>&1; echo "profile one_sided rc=$?"
