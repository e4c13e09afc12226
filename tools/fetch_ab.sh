#!/bin/bash
# FETCH_SIZE + duration of the pair-once kernel for the two tile launch orders (on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ORDER in 0 1; do
  export NBODY_SYM_TILE_ORDER=$ORDER
  OUT=$REPO/gpurun_out/fetch_order$ORDER
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -- python3 $REPO/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 0 > "$OUT.log" 2>&1
  python3 - "$OUT" $ORDER <<'PY'
import csv, glob, sys
f = [float(r["Counter_Value"]) for p in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") for r in csv.DictReader(open(p)) if "force_sym_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for p in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv") for r in csv.DictReader(open(p)) if "force_sym_kernel" in r["Kernel_Name"]]
print("order", sys.argv[2], "FETCH GB (x2 corrected)", sum(f) / len(f) * 1024 * 2 / 1e9, "ms", sum(d) / len(d))
PY
done
