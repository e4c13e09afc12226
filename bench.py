#!/usr/bin/env python3
"""Headline benchmark: body-body interactions/sec of the all-pairs step at N = 2^20 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment, bench.py starts the N ranks itself (a torch.distributed.run child,
before this process touches a GPU) and relays rank 0's line; it exits non-zero if fewer than N devices or ranks come up.

A "step" is one pass of the hot path (the forces of all N x N ordered body-body interactions + kick-drift
update) over one synthetic Plummer-sphere state resident in HBM.  Rank 0 prints ONE JSON line.

* value     = N^2 * K / wall time of K steps (max over ranks, barrier + synchronize on both sides).
* --force-mode pair_once (default): nbody::force_sym_kernel evaluates each unordered pair once and applies it to
              both bodies (the reference's own VERSION 3 idea, kernel.cu:703-774, without atomics); one_sided:
              nbody::force_kernel_r4 evaluates every ordered interaction.  Both on 1..8 GPUs; at N = 1 the other mode
              is timed too and reported in "other_force_mode".
* roofline  = the dominant force kernel against the fp32 vector peak: 20 flop x the pair evaluations one
              launch EXECUTES / its HIP-event duration, measured in this run on the stream the kernel runs on.  For the
              pair-once kernel that is half the ordered interactions it accounts for -- no 2x credit;
              the flop its instructions really perform (26 per unordered pair) and the credited reading are reported
              under their own names.  "traffic" is the HBM bytes per launch from the committed rocprofv3 PMC summary
              when one exists for this kernel and size, else null.
* cpu_baseline = the CPU oracle's scalar all-pairs loop (a port; the reference has no CPU path), timed on this
              host's cores on a row slab of the same workload (N = 1 run only).
Multi-GPU: total N is fixed, rows are sharded over the ranks => "scaling": "strong".  With the nccl backend (the product
path) every per-step exchange runs inside the library (nbody_multi_*, csrc/nbody_multi.hip: RCCL all-gather or ring);
torch.distributed carries the RCCL id at start-up, the barriers and the max over ranks of the elapsed time.  The line then
carries "per_rank" (every rank's kernels, exchange times and host enqueue time per step, from the library's own events) and
"peer_copy_leg" (the same job from one process with hipMemcpyPeerAsync instead of RCCL, run in a child process after the
ranks are done).  A rank whose exchange does not arrive within --exchange-timeout seconds (60) prints ONE JSON line with
"error" and exits non-zero.  --transport peer_copy runs that secondary measurement as the main one (no launcher, no RCCL).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_INTERACTION = 20.0        # SURVEY.md 8d (GPU Gems 3 ch.31 convention)
# flop the instructions of one pair evaluation really perform (= what rocprofv3's SQ_INSTS_VALU_*_F32 counters add up to)
FLOP_INSTRUCTIONS = {"one_sided": 19.0,   # 3 sub, 6 fma, 3 mul, 1 rsq
                     "pair_once": 26.0,   # 3 sub, 9 fma, 4 mul, 1 rsq, for the two interactions of the pair
                     # splits whose bodies share one mass (every split of the benchmark's equal-mass sphere): the mass
                     # multiplies leave the inner loop (nbody_set_equal_mass_path, include/nbody.h)
                     "one_sided_equal_mass": 18.0, "pair_once_equal_mass": 24.0}
KERNEL_NAME = {"one_sided": "nbody::force_kernel_r4pk", "pair_once": "nbody::force_sym_kernel"}
PEAK_FP32_VECTOR_TFLOPS = 157.3    # MI355X_MICROARCH.md, chip-level parameters: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(pos, softening, target_seconds):
    """Scalar all-pairs rows of the oracle on all host cores, on a row slab sized to ~target_seconds."""
    import oracle
    oracle.build()
    visible, quota = oracle.cpus_visible(), oracle.cpu_quota_cores()
    cores = oracle.host_threads()                 # the GPU box shows 256 CPUs and grants 16 cores' worth of time
    n = pos.shape[0]
    rows = min(n, 8 * cores)
    t = time.perf_counter()
    oracle.accel_f32(pos, 0, rows, 0, n, softening, threads=cores)
    probe = time.perf_counter() - t
    rows = int(min(n, max(cores, rows * target_seconds / max(probe, 1e-6))))
    rows -= rows % cores or 0
    rows = max(rows, cores)
    t = time.perf_counter()
    oracle.accel_f32(pos, 0, rows, 0, n, softening, threads=cores)
    dt = time.perf_counter() - t
    rows1 = max(1, int(rows / cores / 8))        # about an eighth of the all-core time, on one core
    t = time.perf_counter()
    oracle.accel_f32(pos, 0, rows1, 0, n, softening, threads=1)
    dt1 = time.perf_counter() - t
    return {"value": rows * n / dt, "unit": "interactions/s", "cores": cores, "kind": "port",
            "value_one_core": rows1 * n / dt1, "cpus_visible": visible, "cpu_quota_cores": quota,
            "sample": f"rows [0,{rows}) x all {n} columns of the same state, {dt:.1f} s, "
                      f"oracle/nbody_oracle.c reference-order fp32 (extrapolates linearly in rows)"}


def reference_size_leg(nb):
    """The only timing the reference publishes: "1.6 ms" per step for its final VERSION 3 on an RTX 4090
    (kernel.cu:73), most plausibly on its default-able dataset 0, galaxy_20K.bin.  The same
    input here -- the file itself (tests/golden/galaxy_20K.bin, load_data(0), kernel.cu:975-981), padded the reference's
    way to 20225 bodies (:260-278), its dt and effective softening; side information, not the headline metric."""
    import torch
    from n_body_problem_amd import datasets
    path = os.path.join(ROOT, "tests", "golden", "galaxy_20K.bin")
    pos, vel = datasets.read_tipsy(path)
    n = pos.shape[0]
    ppos, pvel = nb.pad_reference_style(pos, vel)            # the reference's 20225-body buffers
    s = nb.initialize(ppos.shape[0], force_mode="auto")     # nbody_create_auto: what INTEGRATION.md section 2 patches in
    mode, split_len = s.force_mode, s.split_len
    s.setParticlesPosition(ppos)
    s.setParticlesVelocity(pvel)
    s.step_n(20, nb.TIME_TICK, nb.SOFTENING_VERSION3)
    torch.cuda.synchronize()
    k = 200
    runs = []
    for _ in range(5):   # the best of five runs of k steps (the clock needs a moment to settle on launches this short)
        t0 = time.perf_counter()
        s.step_n(k, nb.TIME_TICK, nb.SOFTENING_VERSION3)
        torch.cuda.synchronize()
        runs.append(1e3 * (time.perf_counter() - t0) / k)
    ms = min(runs)
    s.close()
    return {"input": "tests/golden/galaxy_20K.bin (the reference's data/galaxy_20K.bin)", "n_bodies": n,
            "n_padded": int(ppos.shape[0]), "force_mode": mode, "split_len": split_len, "ms_per_step": ms,
            "ms_per_step_runs": runs, "steps_per_run": k, "interactions_per_s": float(n) * n / (ms * 1e-3),
            "reference_ms_per_step": 1.6, "reference_hardware": "RTX 4090 (source comment kernel.cu:73, N inferred)",
            "speedup_vs_reference_comment": 1.6 / ms}


def executed_pairs(mode, n, split_len, rows_here):
    """Pair evaluations the TIMED force launches of rank 0 execute per step.  Pair-once mode: the S (S - 1) / 2 tiles of two
    different splits, and the S diagonal tiles where the tile launch serves them (as full squares: every ordered pair of the
    split, 0.4 % of the launch's evaluations at N = 2^20); elsewhere they run in their own kernel on the auxiliary stream, are
    not in `force_ms` and therefore not counted here either."""
    if mode == "one_sided":
        return rows_here * n
    S = -(-n // split_len)
    tiles = S * (S - 1) / 2
    # the tile launch serves the diagonal tiles too, as full squares that keep their row side: small systems
    # (force_sym_quarter_kernel) and, since round 4, strips (2048-body splits, a split count that is a multiple of 32)
    if split_len in (256, 512) or (split_len == 2048 and S % 32 == 0):
        tiles += S
    return tiles * split_len * split_len * rows_here / n


def roofline(mode, n, split_len, rows_here, steps, tm, equal_mass=True):
    """the fraction is computed from EXECUTED pair evaluations x 20 flop, never from the 2x credit a
    pair-once kernel could claim for the ordered interactions it accounts for.  Both other readings are reported beside
    it under their own names."""
    force_s = max(tm["force_ms"] / 1e3, 1e-12)
    launches = max(tm["force_launches"], 1)
    executed = executed_pairs(mode, n, split_len, rows_here) * steps
    achieved = FLOP_PER_INTERACTION * executed / force_s / 1e12
    instr_flop = FLOP_INSTRUCTIONS[mode + ("_equal_mass" if equal_mass else "")] * executed / force_s / 1e12
    credit = FLOP_PER_INTERACTION * rows_here * n * steps / force_s / 1e12
    traffic = committed_traffic(n, KERNEL_NAME[mode]) if rows_here == n else None
    return {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
            "traffic": traffic[0] if traffic else None,
            "traffic_source": ("committed rocprofv3 PMC summary profiles/" + traffic[1] + " (not measured in this run)") if traffic else None,
            "kernel": KERNEL_NAME[mode],
            "flop_per_launch": FLOP_PER_INTERACTION * executed / launches,
            "executed_pair_evaluations_per_s": executed / force_s,
            "ordered_interactions_accounted_per_s": rows_here * n * steps / force_s,
            "frac_instruction_flop": instr_flop / PEAK_FP32_VECTOR_TFLOPS,
            "frac_if_credited_per_ordered_interaction": credit / PEAK_FP32_VECTOR_TFLOPS,
            "avg_launch_ms": tm["force_ms"] / launches, "launches": tm["force_launches"],
            "inner_loop": ("equal-mass splits: the mass leaves the loop, " + ("14 + 1" if mode == "pair_once" else "11 + 1")
                           if equal_mass else "general masses, " + ("16 + 1" if mode == "pair_once" else "12 + 1")) +
                          " fp32 operations per pair evaluation, issued as packed instructions (two " +
                          ("columns" if mode == "pair_once" else "rows") + " each)",
            "note": "fp32 vector (VALU) peak = fp32 MFMA dense peak = 157.3 TFLOP/s; no MFMA used.  achieved/frac: executed "
                    "pair evaluations x 20 flop (SURVEY.md 8d).  frac_instruction_flop: the flop the kernel's instructions "
                    "really perform (general masses: one_sided 19 per evaluation, pair_once 26 = 3 sub, 9 fma, 4 mul, 1 rsq "
                    "for TWO interactions; equal-mass splits: 18 and 24).  frac_if_credited_per_ordered_interaction: 20 flop "
                    "x the ordered interactions the launch accounts for -- NOT a utilisation figure for the pair-once "
                    "kernel.  Rank 0's kernels."}


def other_mode_leg(nb, mode, n, pos, vel, args, equal_mass_path=True, body_order=None):
    """Another kernel on the same state and GPU (N = 1 run only): the force mode that is NOT the headline, the headline
    mode with the equal-mass inner loop switched off (every split down the general-mass path), or the headline mode with
    the bodies left in the order the generator produced them."""
    import torch
    s = nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0,
                       body_order=body_order or args.body_order)
    s.set_force_mode(mode)
    s.set_equal_mass_path(equal_mass_path)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.timing(True)
    s.step(args.dt, args.softening)
    s.read_timing()
    torch.cuda.synchronize()
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step(args.dt, args.softening, sync=False)
    s.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = s.read_timing()
    out = {"force_mode": mode, "body_order": s.body_order, "value": float(n) * n * steps / dt, "unit": "interactions/s",
           "ms_per_step": 1e3 * dt / steps, "steps": steps, "roofline": roofline(mode, n, s.split_len, n, steps, tm, equal_mass_path)}
    s.close()
    return out


def committed_traffic(n, kernel):
    """HBM bytes per launch from a committed PMC summary (profiles/*pmc*.json) for this kernel and body count, if any."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("n") == n and "hbm_bytes_per_force_launch" in d and d.get("kernel", "").split("<")[0] in kernel:
            best = (d["hbm_bytes_per_force_launch"], os.path.basename(f))
    return best


def forwarded_args(argv):
    """The command line as the rank processes get it behind torch.distributed.run: argparse checks every "--x" on the line
    against the launcher's own options before it reaches the script's, and "--n" is an ambiguous prefix there ("--nnodes",
    "--nproc-per-node", ...); its alias "--bodies" is not."""
    return ["--bodies" if a == "--n" else "--bodies=" + a[4:] if a.startswith("--n=") else a for a in argv]


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising a HIP runtime here (on ROCm builds without amdsmi
    torch.cuda.device_count() falls back to hipGetDeviceCount, which does): the visibility variables, else the kfd topology
    (a node with SIMDs is a GPU).  None when neither says anything -- the ranks check for themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
            count += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            return None
    return count


def launch_ranks(args):
    """--gpus N without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU) from a parent that
    never touches a GPU (it does not even count them through the HIP runtime), relay their output and return the exit
    code -- non-zero when fewer than N devices are visible, a rank fails, or rank 0's line does not report n_gpus = N."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if not args.single_device and have is not None and have < args.gpus:
        log(f"bench.py: --gpus {args.gpus} but {have} device(s) visible")
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    forwarded = forwarded_args(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *forwarded]
    log("bench.py: launching", " ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line, errors = None, []
    for out in res.stdout.splitlines():
        if out.startswith("{"):
            try:
                d = json.loads(out)
                if "metric" in d and "error" in d:
                    errors.append(out)
                    continue
                if "metric" in d:
                    line = out
                    continue
            except ValueError:
                pass
        log(out)
    if res.returncode != 0:
        log(f"bench.py: the ranks exited with code {res.returncode}")
        if errors:
            print(errors[0], flush=True)              # the evidence a failed exchange leaves (one line, like a result)
        return res.returncode
    if line is None or json.loads(line).get("n_gpus") != args.gpus:
        log("bench.py: rank 0 printed no result line for the requested rank count")
        return 3
    print(line, flush=True)
    return 0


def error_line(args, rank, world, what):
    """A failed run leaves ONE JSON line with "error" on stdout (and the caller exits non-zero): a stalled first contact of
    the exchange must leave evidence, not a kill at the driver's limit."""
    return json.dumps({"metric": "body-body interactions/sec", "value": None, "unit": "interactions/s", "n_gpus": world,
                       "steps": args.steps, "warmup": args.warmup, "error": what, "rank": rank,
                       "config": {"n_bodies": args.n, "transport": args.transport, "exchange": args.exchange,
                                  "exchange_timeout_s": args.exchange_timeout}})


def per_step(tm, steps):
    """One rank's share of a step from the library's event totals (milliseconds per step; sums of durations, not wall)."""
    k = max(1, steps)
    out = {"force_ms": tm["force_ms"] / k, "force_launches_per_step": tm["force_launches"] / k,
           "update_ms": tm["update_ms"] / k, "aux_ms": tm.get("aux_ms", 0.0) / k}
    for key in ("host_enqueue_ms", "pos_exchange_comm_ms", "pos_exchange_wait_ms", "column_sum_exchange_ms", "reorder_ms"):
        if key in tm:
            out[key] = tm[key] / k
    return out


def peer_copy_leg(args, timeout_s=120.0):
    """The same job with the ranks' exchange done by peer copies from ONE process (nbody_multi_create over N devices,
    hipMemcpyPeerAsync instead of RCCL): a labelled secondary measurement, run in a child process after the ranks are
    done, so that a scaling curve exists even if the RCCL leg misbehaves -- and so that nothing it does can lose the main
    line."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID") and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--bodies", str(args.n), "--dt", str(args.dt), "--softening", str(args.softening), "--exchange",
           args.exchange, "--force-mode", args.force_mode, "--body-order", args.body_order, "--transport", "peer_copy",
           "--no-cpu-baseline", "--no-extra-legs", "--no-sanity"] + (["--single-device"] if args.single_device else [])
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s:.0f} s"}
    for out in res.stdout.splitlines():
        if out.startswith("{"):
            try:
                d = json.loads(out)
            except ValueError:
                continue
            keep = ("value", "ms_per_step", "n_gpus", "error", "per_rank", "roofline", "force_only_interactions_per_s")
            leg = {k: d[k] for k in keep if k in d}
            leg["transport"] = "peer_copy: one process drives all ranks, hipMemcpyPeerAsync between the replicas (no RCCL)"
            return leg
    return {"error": f"exit code {res.returncode}: " + res.stderr[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--bodies", dest="n", type=int, default=1 << 20, help="bodies (BASELINE.json: 2^20)")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--softening", type=float, default=1e-3)
    ap.add_argument("--exchange", default=os.environ.get("NBODY_EXCHANGE", "allgather"), choices=["allgather", "ring"])
    ap.add_argument("--transport", default=os.environ.get("NBODY_TRANSPORT", "rccl"), choices=["rccl", "peer_copy"],
                    help="rccl (the product path: one process per GPU, RCCL over xGMI); peer_copy: ONE process drives all "
                         "--gpus ranks and moves the rows with hipMemcpyPeerAsync (no launcher, no RCCL) -- a labelled "
                         "secondary measurement; an rccl run with --gpus > 1 appends it as \"peer_copy_leg\"")
    ap.add_argument("--exchange-timeout", type=float, default=60.0,
                    help="seconds a rank waits for an exchange before it aborts the communicators and reports (library "
                         "default: 600)")
    ap.add_argument("--rows-per-lane", type=int, default=0)
    ap.add_argument("--body-order", default=os.environ.get("NBODY_BODY_ORDER", "morton"), choices=["morton", "given"],
                    help="how the library stores the bodies it is given: along a Morton curve (laid on the device at upload, "
                         "undone at download: neighbours in memory are neighbours in space, fewer operand bits toggle, the "
                         "power-limited clock rises) or in the generator's (random) order; the other one is reported as a leg")
    ap.add_argument("--reorder-every", type=int, default=0,
                    help="body-order morton: refresh the layout on the device every so many steps (0: never)")
    ap.add_argument("--force-mode", default=os.environ.get("NBODY_FORCE_MODE", "pair_once"),
                    choices=["pair_once", "one_sided", "symmetric"],
                    help="pair_once (= symmetric): each unordered pair once; one_sided: every ordered interaction")
    ap.add_argument("--no-equal-mass-path", action="store_true",
                    help="send every split down the general-mass inner loops (what a body set with arbitrary masses gets; the "
                         "N = 1 run reports it as the general_mass_path leg anyway -- this makes it the main measurement, for profiling)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sanity", action="store_true", help="skip the energy / replica checks around the timed region")
    ap.add_argument("--no-extra-legs", "--no-pair-once", dest="no_extra_legs", action="store_true",
                    help="skip the extra legs (the other force mode, the reference's N = 20000, the peer-copy leg)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the torch.distributed process group, which only carries the RCCL id and the final "
                         "reductions of this script (gloo = NBODY_RENDEZVOUS=gloo: rehearsals with several ranks on one GPU); "
                         "every per-step exchange is the library's own RCCL call")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()
    mode = "pair_once" if args.force_mode == "symmetric" else args.force_mode
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    one_process = args.transport == "peer_copy" or args.gpus == 1
    if not one_process and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))           # before anything here touches a GPU
    world = args.gpus if one_process else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if one_process else int(os.environ.get("RANK", "0"))
    local_rank = 0 if one_process else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure a different job")
    if mode == "pair_once" and 8 % world:
        log("the pair-once mode shards over 1, 2, 4 or 8 ranks: falling back to --force-mode one_sided")
        mode = "one_sided"

    import torch
    import torch.distributed as dist
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.single_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} devices")
    torch.cuda.set_device(local_rank)
    distributed = world > 1 and not one_process
    # NBODY_RENDEZVOUS=gloo: the product branch below (library-owned exchange, one rank per process) with the process
    # group on gloo -- for rehearsing it where real RCCL cannot run (several ranks on one GPU, the library linked with the
    # RCCL test double of tests/fake_rccl: NBODY_AMD_LIBRARY).  The driver's runs never set it.
    gloo_rendezvous = os.environ.get("NBODY_RENDEZVOUS") == "gloo" or args.backend == "gloo"
    if distributed:
        if not gloo_rendezvous:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    n = args.n
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    try:
        if world == 1:
            system = nb.NBodySystem(n, device=local_rank, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0,
                                    body_order=args.body_order)
            system.set_force_mode(mode)
            kernels = system
        elif one_process:
            # every rank in this process, rows moved by peer copies: no launcher, no RCCL (the labelled secondary measurement)
            devices = [0] * world if args.single_device else list(range(world))
            system = MultiGpuSystem(n, devices=devices, force_mode=mode, exchange=args.exchange, transport="peer_copy",
                                    body_order=args.body_order)
            kernels = system.kernels
        else:
            # one rank per process, the exchange inside the library (the process group only carries the RCCL id)
            system = MultiGpuSystem.from_torch_distributed(n, local_rank, exchange=args.exchange, force_mode=mode,
                                                           body_order=args.body_order, create_timeout=args.exchange_timeout)
            kernels = system.kernels
        library_exchange = isinstance(system, MultiGpuSystem)
        rccl_ranks = system.info()["rccl_ranks"] if library_exchange else None
        if library_exchange and not one_process and rccl_ranks != world:
            raise SystemExit(f"bench.py: the RCCL communicator has {rccl_ranks} ranks, expected {world}")
        if library_exchange:
            system.set_timeout(args.exchange_timeout)   # well inside any driver limit: a dead peer is reported, not waited for
        if args.reorder_every and hasattr(system, "set_reorder_period"):
            system.set_reorder_period(args.reorder_every)
        kernels.set_rows_per_lane(args.rows_per_lane)
        if args.no_equal_mass_path:
            for k in ([system.shard(i) for i in range(system.local_ranks)] if library_exchange else [kernels]):
                k.set_equal_mass_path(False)
        system.setParticlesPosition(pos)
        system.setParticlesVelocity(vel)
        info = kernels.device_info()

        def barrier():
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()

        def run_steps(k):
            if args.reorder_every and not library_exchange:
                system.step_n(k, args.dt, args.softening)       # the Python layer's schedule of refreshes
                return
            for _ in range(k):
                system.step(args.dt, args.softening, sync=False)
            system.sync()

        def read_timing():
            if library_exchange:                                 # kernels + exchanges, every local rank
                return [system.read_timing(i) for i in range(system.local_ranks)]
            return [kernels.read_timing()]

        # outside the timed region: the energy before and after, and a check that every rank ends with the same positions
        e_before = None if args.no_sanity else system.energy(args.softening)
        (system if library_exchange else kernels).timing(True)
        run_steps(args.warmup)
        read_timing()  # reset the event totals

        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo_rendezvous else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        local_tm = read_timing()
        tm = local_tm[0]
        (system if library_exchange else kernels).timing(False)
        per_rank = [dict(rank=rank + i, **per_step(t, args.steps)) for i, t in enumerate(local_tm)]
        if distributed:
            gathered = [None] * world
            dist.all_gather_object(gathered, per_rank)
            per_rank = [x for part in gathered for x in part]
        sanity = None
        if not args.no_sanity:
            e_after = system.energy(args.softening)
            if library_exchange:
                identical = system.replicas_identical()
            else:
                digest = system.positions.view(torch.int32).to(torch.int64).sum().reshape(1)
                lo, hi = digest.clone(), digest.clone()
                if distributed:
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                identical = int(lo.item()) == int(hi.item())
            sanity = {"energy_before": float(e_before[2]), "energy_after": float(e_after[2]),
                      "dE_over_E0": float((e_after[2] - e_before[2]) / abs(e_before[2])),
                      "steps_between": args.warmup + args.steps,
                      "position_replicas_identical_on_all_ranks": bool(identical)}
    except nb.NBodyError as e:
        # the library gave up on an exchange (timeout, RCCL error) or refused the job: say so in one line and leave
        print(error_line(args, rank, world, str(e)), flush=True)
        log(f"bench.py: rank {rank}: {e}")
        os._exit(1)                                     # no barrier, no destructor may wait for a peer that is gone

    out = None
    if rank == 0:
        interactions = float(n) * float(n) * args.steps
        rows_here = n / world
        force_s = max(tm["force_ms"] / 1e3, 1e-12)
        out = {
            "metric": "body-body interactions/sec",
            "value": interactions / elapsed,
            "unit": "interactions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"all-pairs step, N={n} Plummer sphere (BASELINE.json configs[2]), fp32, "
                                   f"softening={args.softening}, dt={args.dt}",
                       "n_bodies": n, "parallelism": f"rows sharded x{world}" if world > 1 else "1 GPU",
                       "exchange": args.exchange if world > 1 else None,
                       "backend": ("peer_copy (one process, hipMemcpyPeerAsync; secondary measurement)" if one_process else
                                   "rccl calls of the loaded library, process group on gloo (rehearsal)" if gloo_rendezvous else
                                   "rccl") if world > 1 else None,
                       "exchange_owner": ("library (nbody_multi_*, csrc/nbody_multi.hip)" if library_exchange else
                                          "torch.distributed rehearsal harness") if world > 1 else None,
                       "exchange_timeout_s": args.exchange_timeout if library_exchange else None,
                       "rccl_ranks": rccl_ranks,
                       "split_len": int(getattr(system, "split_len", 0)), "seed": nb.CONFIG_SEED[3],
                       "force_mode": mode, "equal_mass_path": not args.no_equal_mass_path,
                       # the layout the library keeps the generator's bodies in (laid on the device at upload, undone at
                       # download; the other order is the `other_body_order` leg of the N = 1 run)
                       "body_order": getattr(system, "body_order", "given"),
                       "reorder_every": args.reorder_every or None},
            "roofline": roofline(mode, n, int(getattr(system, "split_len", 0)), rows_here, args.steps, tm,
                                 equal_mass=bool(np.all(pos[:, 3] == pos[0, 3])) and not args.no_equal_mass_path),
            "force_only_interactions_per_s": rows_here * n * args.steps / force_s * world,
            "update_ms_per_step": tm["update_ms"] / args.steps,
            "overlapped_aux_ms_per_step": tm.get("aux_ms", 0.0) / args.steps,   # diagonal tiles + early summation, beside the tiles
            # where every rank's step goes (library event totals, ms per step): its kernels, the position exchange on the
            # communication stream and as the waiting force launch saw it, the pair-once column-sum exchange (not hidden,
            # by design), and the host time spent enqueuing a step
            "per_rank": per_rank if world > 1 else None,
            # pair-once mode on one context: the tile launches per pass and what the two partial-sum arrays hold (two of the
            # parts; n^2 / split_len 12-byte entries would be the whole pass)
            "summation_parts": tm["force_launches"] // max(1, args.steps) if mode == "pair_once" and world == 1 else None,
            "partial_sum_bytes": system.partial_sum_bytes() if hasattr(system, "partial_sum_bytes") else None,
            "device": info,
            "sanity": sanity,
        }
        if world == 1 and not args.no_extra_legs:
            out["other_force_mode"] = other_mode_leg(nb, "one_sided" if mode == "pair_once" else "pair_once", n, pos, vel, args)
            out["general_mass_path"] = other_mode_leg(nb, mode, n, pos, vel, args, equal_mass_path=False)
            out["other_body_order"] = other_mode_leg(nb, mode, n, pos, vel, args,
                                                     body_order="given" if args.body_order == "morton" else "morton")
            out["reference_size"] = reference_size_leg(nb)
            both = {mode: out["roofline"], out["other_force_mode"]["force_mode"]: out["other_force_mode"]["roofline"]}
            out["north_star_target"] = {
                "target": ">= 0.40 of the fp32 peak on the force kernel at N = 2^20 (BASELINE.json)",
                "one_sided_kernel_frac": both["one_sided"]["frac"],
                "pair_once_kernel_frac_20_flop_per_executed_evaluation": both["pair_once"]["frac"],
                "pair_once_kernel_frac_instruction_flop": both["pair_once"]["frac_instruction_flop"],
                "note": "the pair-once kernel is the default because it delivers more interactions per second; it gets no "
                        "credit for the second body of a pair (SURVEY.md 8d), so its fraction is lower than the one-sided "
                        "kernel's, which is the kernel the target was written for"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, args.softening, args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    system.close()
    if out is not None and distributed and not gloo_rendezvous and not args.no_extra_legs:
        out["peer_copy_leg"] = peer_copy_leg(args)      # a child process, after this rank has let go of its GPU objects
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
