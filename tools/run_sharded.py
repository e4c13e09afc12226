#!/usr/bin/env python3
"""Long sharded run with energy reports (BASELINE.json configs[4]: N = 4 194 304 on 8 GPUs, 1000 steps, energy drift
and interactions/s).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_sharded.py \
        --bodies 4194304 --steps 1000 --energy-every 100
    python tools/run_sharded.py --bodies 131072 --steps 1000 --energy-every 100          # one GPU

Rank 0 prints one line per report and a final JSON summary.  Each energy report is an O(N^2) potential pass of its own
(one row per lane: about the cost of two force passes at N = 2^22), outside the timed steps."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=1 << 22)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--softening", type=float, default=1e-2)
    ap.add_argument("--energy-every", type=int, default=100)
    ap.add_argument("--force-mode", default="pair_once", choices=["pair_once", "one_sided"])
    ap.add_argument("--integrator", default="kick_drift", choices=["kick_drift", "kdk"])
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "ring"])
    ap.add_argument("--split-len", type=int, default=0, help="0 = the mode's default for this body count")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--body-order", default="given", choices=["given", "morton"],
                    help="morton: the library stores the bodies along a Morton curve (3-4 %% more clock); nccl backend only")
    ap.add_argument("--reorder-every", type=int, default=0, help="with --body-order morton: refresh the layout every so many steps")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import sharded_system
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    # NBODY_RENDEZVOUS=gloo: the library-owned exchange with the process group on gloo -- rehearsal on one GPU with the RCCL
    # test double of tests/fake_rccl (NBODY_AMD_LIBRARY), as in bench.py
    gloo_rendezvous = os.environ.get("NBODY_RENDEZVOUS") == "gloo"
    if world > 1:
        if args.backend == "nccl" and not gloo_rendezvous:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    n = args.bodies
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[5])
    # nccl backend (or a single process): the exchange runs inside the library (nbody_multi_*); gloo: the rehearsal harness
    if args.backend == "nccl":
        from n_body_problem_amd.multi import MultiGpuSystem
        s = MultiGpuSystem.from_torch_distributed(n, local, exchange=args.exchange, force_mode=args.force_mode,
                                                  integrator=args.integrator, split_len=args.split_len, body_order=args.body_order)
    else:
        s = sharded_system(n, device=local, exchange=args.exchange, force_mode=args.force_mode, integrator=args.integrator,
                           split_len=args.split_len, body_order=args.body_order)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    if args.reorder_every and hasattr(s, "set_reorder_period"):
        s.set_reorder_period(args.reorder_every)
    del pos, vel
    e0 = s.energy(args.softening)
    if rank == 0:
        print(f"N={n} ranks={world} force_mode={args.force_mode} integrator={args.integrator} dt={args.dt} "
              f"softening={args.softening}  E0={e0[2]:.9e} (K {e0[0]:.6e} U {e0[1]:.6e})", flush=True)
    done, busy, worst = 0, 0.0, 0.0
    every = args.energy_every if args.energy_every > 0 else args.steps
    while done < args.steps:
        k = min(every, args.steps - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.step_n(k, args.dt, args.softening)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        busy += dt
        done += k
        e = s.energy(args.softening)
        drift = (e[2] - e0[2]) / abs(e0[2])
        worst = max(worst, abs(drift))
        if rank == 0:
            print(f"step {done:6d}  E={e[2]:.9e}  dE/E0={drift:+.3e}  {1e3 * dt / k:8.3f} ms/step  "
                  f"{float(n) * n * k / dt:.3e} interactions/s", flush=True)
    mom = s.momentum()
    if rank == 0:
        print(json.dumps({"n_bodies": n, "n_gpus": world, "steps": args.steps, "force_mode": args.force_mode,
                          "integrator": args.integrator, "dt": args.dt, "softening": args.softening,
                          "interactions_per_s": float(n) * n * args.steps / busy, "ms_per_step": 1e3 * busy / args.steps,
                          "max_abs_dE_over_E0": worst, "final_dE_over_E0": drift,
                          "momentum": [float(x) for x in mom[:3]]}), flush=True)
    s.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
