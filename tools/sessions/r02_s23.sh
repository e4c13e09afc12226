#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/r02_s23_build.log 2>&1 || { tail -20 gpurun_out/r02_s23_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -k "not config5" > gpurun_out/r02_s23_pytest.log 2>&1
rc=$?; tail -8 gpurun_out/r02_s23_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r02_s23_bench.json 2> gpurun_out/r02_s23_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_s23_bench.json'))
print({k:d[k] for k in ("value","ms_per_step","update_ms_per_step","overlapped_aux_ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"])
PY
