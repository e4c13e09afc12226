#!/bin/bash
# PMC pass over the pair-once kernel (on the GPU box): tools/pmc_sym.sh <tag>
set -u
TAG=${1:-sym}
[ -n "${2:-}" ] && export NBODY_AMD_LIBRARY=$(realpath "$2")
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu-baseline --no-pair-once --force-mode symmetric"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -- python3 $BENCH --steps 2 --warmup 0 > "$OUT/pmc_SQ.log" 2>&1 || echo "pmc SQ failed"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_LDS" -- python3 $BENCH --steps 2 --warmup 0 > "$OUT/pmc_LDS.log" 2>&1 || echo "pmc LDS failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*")):
    if not d.endswith(("SQ", "LDS")): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "force_sym_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Kernel_Name"][:60], r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    dur = []
    for f in glob.glob(d + "/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "force_sym_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(d.split("/")[-1], meta if acc else None, "avg ms", sum(dur) / max(len(dur), 1))
    for k, v in sorted(acc.items()):
        print("  %-26s %.6g" % (k, sum(v) / len(v)))
PY
