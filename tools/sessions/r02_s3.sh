#!/bin/bash
# round 2, GPU session 3: full GPU suite + bench + rocprofv3 kernel trace of the bench command.
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s3_build.log 2>&1 || { tail -20 gpurun_out/r02_s3_build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r02_s3_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r02_s3_pytest.log; echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r02_s3_bench.json 2> gpurun_out/r02_s3_bench.err
rc=$?; tail -c 600 gpurun_out/r02_s3_bench.json; echo "bench rc=$rc"
