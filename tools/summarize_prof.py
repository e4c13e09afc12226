#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (written by tools/profile.sh on the GPU box) into the
tracked summaries under profiles/:  <round>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats) and
<round>_pmc[_<mode>]_n<N>.json (per-launch PMC counters of the dominant force kernel with the gfx950 corrections of
MI355X_MICROARCH.md: FETCH_SIZE is in KiB and reads HALF the bytes of a 16-B/lane coalesced stream).

    python tools/summarize_prof.py <tag> <round> [N] [kernel substring: force_kernel_r4 | force_sym_kernel] [mode tag]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(paths):
    """Of the files several profiler runs left in one directory (named <pid>_*.csv), those of the latest run."""
    paths = list(paths)
    if not paths:
        return []
    latest = max(paths, key=os.path.getmtime)
    pid = os.path.basename(latest).split("_")[0]
    return [f for f in paths if os.path.basename(f).split("_")[0] == pid and os.path.dirname(f) == os.path.dirname(latest)]


def main(tag, rnd, n, kern="force_kernel", mode=""):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{rnd}_kernel_stats{'_' + mode if mode else ''}.csv"))
    out = {"n": n, "round": rnd, "kernel": "nbody::" + kern, "source": f"tools/profile.sh {tag} (rocprofv3, separate --pmc passes)"}
    counters = defaultdict(list)
    durations = []
    vgpr = None
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for f in newest(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    vgpr = r.get("VGPR_Count")
                    out["kernel_name"] = r["Kernel_Name"]
                    out["lds_block_size"] = int(r["LDS_Block_Size"])
    steps = 0
    for f in newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                durations.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            if "update_kernel" in r["Kernel_Name"] or "update_sym_kernel" in r["Kernel_Name"]:   # one per step
                steps += 1
    # launches of the force kernel per force pass (round 2: the pair-once mode launches its tiles in two parts)
    per_pass = max(1, round(len(durations) / steps)) if steps else 1
    out["force_launches_per_pass"] = per_pass
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    out["counters_per_launch"] = mean
    out["avg_launch_ms_kernel_trace"] = sum(durations) / len(durations) if durations else None
    out["force_pass_ms_kernel_trace"] = out["avg_launch_ms_kernel_trace"] * per_pass if durations else None
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch = mean["FETCH_SIZE"] * 1024.0 * 2.0   # KiB -> B, x2: gfx950 counts 128-B requests as 64 B
        write = mean["WRITE_SIZE"] * 1024.0
        out["hbm_read_bytes_per_force_launch"] = fetch
        out["hbm_write_bytes_per_force_launch"] = write
        out["hbm_bytes_per_force_launch"] = fetch + write
        out["hbm_bytes_per_force_pass"] = (fetch + write) * per_pass
        if durations:
            out["hbm_GBps"] = (fetch + write) / (out["avg_launch_ms_kernel_trace"] * 1e-3) / 1e9
    if "GRBM_GUI_ACTIVE" in mean and durations:
        cyc = mean["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
        out["effective_clock_GHz"] = cyc / (out["avg_launch_ms_kernel_trace"] * 1e-3) / 1e9
        if "SQ_ACTIVE_INST_VALU" in mean:
            # wave-cycles with a VALU instruction in flight per SIMD cycle (quad-cycle counter; > 1 when several waves
            # of a SIMD overlap): not a utilisation fraction
            out["valu_active_wave_cycles_per_simd_cycle"] = mean["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
        if "SQ_WAVE_CYCLES" in mean:
            out["mean_waves_per_simd"] = mean["SQ_WAVE_CYCLES"] * 4.0 / (cyc * 1024.0)
        if "SQ_INSTS_VALU" in mean:
            inter = float(n) * float(n) / 64.0 / per_pass   # wave64 instructions' worth of ORDERED interactions per launch
            out["valu_instructions_per_interaction"] = mean["SQ_INSTS_VALU"] / inter
            out["simd_cycles_per_interaction"] = cyc * 1024.0 / inter
    if all(k in mean for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")) and durations:
        # wave64 instructions x 64 lanes; an FMA is 2 flop, everything else 1 (v_sub_f32 is counted with the adds)
        # force_kernel_r4pk does all its adds, multiplies and FMAs as v_pk_*_f32: the counters see one instruction for two
        # lanes' worth (7.1 VALU instructions per interaction instead of 13.1), so each counts double
        # (round 2: so do the pair-once tiles, two columns per packed instruction)
        pk = 2.0 if ("r4pk" in kern or "force_sym_kernel" in kern) else 1.0
        flop = 64.0 * (pk * (mean["SQ_INSTS_VALU_ADD_F32"] + mean["SQ_INSTS_VALU_MUL_F32"] + 2.0 * mean["SQ_INSTS_VALU_FMA_F32"]) +
                       mean["SQ_INSTS_VALU_TRANS_F32"])
        out["rocprof_flop_per_launch"] = flop
        out["rocprof_TFLOPs"] = flop / (out["avg_launch_ms_kernel_trace"] * 1e-3) / 1e12
        out["rocprof_frac_of_fp32_peak_157.3"] = out["rocprof_TFLOPs"] / 157.3
    out["vgpr_count_reported"] = vgpr
    path = os.path.join(dst, f"{rnd}_pmc{'_' + mode if mode else ''}_n{n}.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20,
         sys.argv[4] if len(sys.argv) > 4 else "force_kernel", sys.argv[5] if len(sys.argv) > 5 else "")
