#!/usr/bin/env python3
"""Headline benchmark: body-body interactions/sec of the all-pairs step at N = 2^20 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (force accumulation over all N x N ordered pairs + kick-drift
update) over one synthetic Plummer-sphere state resident in HBM.  Rank 0 prints ONE JSON line.

* value     = N^2 * K / wall time of K steps (max over ranks, barrier + synchronize on both sides).
* roofline  = the force kernel against the fp32 vector peak: 20 flop per interaction x the
              interactions one launch evaluates / its HIP-event duration, measured in this run on the stream
              the kernel runs on.  The kernel is VALU-bound; "traffic" is the HBM bytes per step from the
              committed rocprofv3 PMC summary when one exists for this size, else null.
* cpu_baseline = the CPU oracle's scalar all-pairs loop (a port; the reference has no CPU path), timed on this
              host's cores on a row slab of the same workload (N = 1 run only).
Multi-GPU: total N is fixed, rows are sharded over the ranks => "scaling": "strong".
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
PEAK_FP32_VECTOR_TFLOPS = 157.3    # MI355X_MICROARCH.md, chip-level parameters: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(pos, softening, target_seconds):
    """Scalar all-pairs rows of the oracle on all host cores, on a row slab sized to ~target_seconds."""
    import oracle
    oracle.build()
    cores = oracle.host_threads()
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
    return {"value": rows * n / dt, "unit": "interactions/s", "cores": cores, "kind": "port",
            "sample": f"rows [0,{rows}) x all {n} columns of the same state, {dt:.1f} s, "
                      f"oracle/nbody_oracle.c reference-order fp32 (extrapolates linearly in rows)"}


def reference_size_leg(nb):
    """The only timing the reference publishes: "1.6 ms" per step for its final VERSION 3 on an RTX 4090
    (kernel.cu:73), most plausibly at galaxy_20K's N = 20000 padded to 20225.  The same
    size here, with the reference's dt and effective softening; side information, not the headline metric."""
    import torch
    n = 20000
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[1])
    ppos, pvel = nb.pad_reference_style(pos, vel)            # the reference's 20225-body buffers
    s = nb.NBodySystem(ppos.shape[0])
    s.setParticlesPosition(ppos)
    s.setParticlesVelocity(pvel)
    s.step_n(20, nb.TIME_TICK, nb.SOFTENING_VERSION3)
    torch.cuda.synchronize()
    k = 200
    t0 = time.perf_counter()
    s.step_n(k, nb.TIME_TICK, nb.SOFTENING_VERSION3)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / k
    s.close()
    return {"n_bodies": n, "n_padded": int(ppos.shape[0]), "ms_per_step": ms, "interactions_per_s": float(n) * n / (ms * 1e-3),
            "reference_ms_per_step": 1.6, "reference_hardware": "RTX 4090 (source comment kernel.cu:73, N inferred)",
            "speedup_vs_reference_comment": 1.6 / ms}


def pair_once_leg(nb, n, pos, vel, args):
    """The experimental pair-once kernel on the same state, reported BESIDE the headline, never as
    it: N^2/t for comparison, and the roofline fraction from the pair evaluations it actually executes."""
    import torch
    try:
        s = nb.NBodySystem(n, split_len=nb.PAIR_ONCE_SPLIT_LEN)
        s.set_force_mode("symmetric")
    except nb.NBodyError as e:
        return {"skipped": str(e)}
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
    L = s.split_len
    S = -(-n // L)
    executed = S * (S + 1) / 2 * L * L * steps
    s.close()
    force_s = max(tm["force_ms"] / 1e3, 1e-12)
    return {"value": float(n) * n * steps / dt, "unit": "interactions/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "kernel": "nbody::force_sym_kernel", "executed_pair_evaluations_per_s": executed / force_s,
            "roofline_frac_executed": FLOP_PER_INTERACTION * executed / force_s / 1e12 / PEAK_FP32_VECTOR_TFLOPS,
            "note": "each unordered pair once, applied to both bodies; single GPU only; agrees with the headline kernel "
                    "to rounding (tests/test_parity_gpu.py), bit-reproducible; NOT the headline value"}


def committed_traffic(n):
    """HBM bytes per step from a committed PMC summary (profiles/*pmc*.json) for this body count, if any."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("n") == n and "hbm_bytes_per_force_launch" in d:
            best = (d["hbm_bytes_per_force_launch"], os.path.basename(f))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--bodies", dest="n", type=int, default=1 << 20, help="bodies (BASELINE.json: 2^20)")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--softening", type=float, default=1e-3)
    ap.add_argument("--exchange", default=os.environ.get("NBODY_EXCHANGE", "allgather"), choices=["allgather", "ring"])
    ap.add_argument("--rows-per-lane", type=int, default=0)
    ap.add_argument("--force-mode", default="one_sided", choices=["one_sided", "symmetric"],
                    help="symmetric = the experimental pair-once kernel (1 GPU only); roofline from EXECUTED pair evaluations")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair-once", action="store_true", help="skip the extra leg that times the experimental pair-once kernel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the product path); gloo only to rehearse the multi-rank flow on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import n_body_problem_amd as nb
    from n_body_problem_amd.sharded import ShardedNBodySystem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    n = args.n
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    if world == 1:
        system = nb.NBodySystem(n, device=local_rank,
                                split_len=nb.PAIR_ONCE_SPLIT_LEN if args.force_mode == "symmetric" else 0)
        kernels = system
    else:
        system = ShardedNBodySystem(n, device=local_rank, exchange=args.exchange)
        kernels = system.kernels
    kernels.set_rows_per_lane(args.rows_per_lane)
    if args.force_mode != "one_sided":
        if world != 1:
            raise SystemExit("--force-mode symmetric is single-GPU")
        kernels.set_force_mode(args.force_mode)
    system.setParticlesPosition(pos)
    system.setParticlesVelocity(vel)
    info = kernels.device_info()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    kernels.timing(True)
    for _ in range(args.warmup):
        system.step(args.dt, args.softening, sync=False)
    system.sync()
    kernels.read_timing()  # reset the event totals

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        system.step(args.dt, args.softening, sync=False)
    system.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = kernels.read_timing()

    out = None
    if rank == 0:
        interactions = float(n) * float(n) * args.steps
        rows_here = n / world
        force_s = tm["force_ms"] / 1e3
        launches = max(tm["force_launches"], 1)
        # executed pair evaluations: N^2 ordered ones, or (S(S+1)/2) x split_len^2 unordered ones in the pair-once mode
        executed = rows_here * n * args.steps
        if args.force_mode == "symmetric":
            L = kernels.split_len
            S = -(-n // L)
            executed = S * (S + 1) / 2 * L * L * args.steps
        flop_per_launch = FLOP_PER_INTERACTION * executed / launches
        achieved = FLOP_PER_INTERACTION * executed / max(force_s, 1e-12) / 1e12
        traffic = committed_traffic(n) if world == 1 and args.force_mode == "one_sided" else None
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
                                   f"softening={args.softening}, dt={args.dt}, LDS tile=256",
                       "n_bodies": n, "parallelism": f"rows sharded x{world}" if world > 1 else "1 GPU",
                       "exchange": args.exchange if world > 1 else None,
                       "backend": ("rccl" if args.backend == "nccl" else args.backend) if world > 1 else None,
                       "split_len": int(getattr(system, "split_len", 0)), "seed": nb.CONFIG_SEED[3],
                       "force_mode": args.force_mode},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": "nbody::force_kernel" if args.force_mode == "one_sided" else "nbody::force_sym_kernel",
                         "executed_pair_evaluations_per_s": executed / max(force_s, 1e-12),
                         "flop_per_launch": flop_per_launch,
                         "avg_launch_ms": tm["force_ms"] / launches, "launches": tm["force_launches"],
                         "note": "fp32 vector (VALU) peak = fp32 MFMA dense peak = 157.3 TFLOP/s; no MFMA used; "
                                 "20 flop per ordered interaction; rank 0's kernels"},
            "force_only_interactions_per_s": rows_here * n * args.steps / max(force_s, 1e-12) * world,
            "update_ms_per_step": tm["update_ms"] / args.steps,
            "device": info,
        }
        if world == 1 and args.force_mode == "one_sided" and not args.no_pair_once:
            out["pair_once"] = pair_once_leg(nb, n, pos, vel, args)
        if world == 1 and args.force_mode == "one_sided" and not args.no_pair_once:
            out["reference_size"] = reference_size_leg(nb)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, args.softening, args.cpu_seconds)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    system.close()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
