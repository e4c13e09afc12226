#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then PMC passes in their
# own runs (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass).  Outputs under gpurun_out/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu-baseline --no-extra-legs --no-sanity $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $BENCH --steps 3 --warmup 1 > "$OUT/trace.log" 2>&1 || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python3 $BENCH --steps 2 --warmup 0 > "$OUT/pmc_$C.log" 2>&1 || echo "pmc $C failed"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -- python3 $BENCH --steps 2 --warmup 0 > "$OUT/pmc_SQ.log" 2>&1 || echo "pmc SQ failed"
# the flop the VALU really executed, by instruction class (FMA counts 2): "rocprof-reported FLOP/s"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d "$OUT/pmc_FLOP" -- python3 $BENCH --steps 2 --warmup 0 > "$OUT/pmc_FLOP.log" 2>&1 || echo "pmc FLOP failed"
find "$OUT" -name "*.csv" | head -50
