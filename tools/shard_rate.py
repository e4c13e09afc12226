"""One rank of P's share of a force pass, timed on one GPU (the other ranks' work is simply not done):
python tools/shard_rate.py [--bodies N] [--world 8] [--rank 3] [--mode pair_once] [--split-len L ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402
from n_body_problem_amd import multi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=1 << 20)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--mode", default="pair_once")
ap.add_argument("--split-len", type=int, nargs="*", default=[2048, 1024])
ap.add_argument("--two-streams", action="store_true")
ap.add_argument("--strip-len", type=int, default=0, help="pair-once: nbody_set_strip_len (0 = the library's rule)")
ap.add_argument("--morton", action="store_true", help="the bodies along the Morton curve (what bench.py's default layout gives every rank)")
args = ap.parse_args()
n = args.bodies
pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
if args.morton:
    perm = nb.morton_order(pos)
    pos, vel = pos[perm].copy(), vel[perm].copy()
for L in args.split_len:
    L = L or (nb.pair_once_split_len(n) if args.mode == "pair_once" else nb.default_split_len(n))
    n_padded, chunk, _ = multi.geometry(n, args.world, args.mode, L)
    assert n_padded == n
    lo = args.rank * chunk
    s = nb.NBodySystem(n, row_lo=lo, row_count=chunk, split_len=L)
    s.set_force_mode(args.mode)
    if args.mode == "pair_once":
        s.set_strip_len(args.strip_len)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel[lo:lo + chunk])
    side = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        ev[0].record()
        s.forces(lo, chunk, 1e-3)
        if args.two_streams:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                s.forces_complement(lo, chunk, 1e-3)
            torch.cuda.current_stream().wait_stream(side)
        else:
            s.forces_complement(lo, chunk, 1e-3)
        if args.mode == "pair_once":
            s.sym_reduce()
        s.update(0.0)
        ev[1].record()
        torch.cuda.synchronize()
        if it:
            best = min(best, ev[0].elapsed_time(ev[1]))
    print(f"N={n} rank {args.rank}/{args.world} mode={args.mode} split_len={L} strip_len={args.strip_len} two_streams={args.two_streams} morton={args.morton}: "
          f"{best:.3f} ms per step share -> x{args.world} ranks = {float(n) * n / best / 1e9:.3f}e12 interactions/s")
    if args.morton:
        # one rank's share of a layout refresh (nbody_multi_reorder without the wire): the permutation from the rank's own replica,
        # the replica gathered in place, the rank's velocity rows gathered out of all rows, the order array composed
        import ctypes
        from n_body_problem_amd._lib import check
        lib, ptr = s._lib, (lambda t: ctypes.c_void_p(t.data_ptr()))
        vel_all = torch.from_numpy(vel).cuda()
        order = torch.arange(n, dtype=torch.int64, device="cuda")
        refresh = 1e9
        for it in range(4):
            torch.cuda.synchronize()
            ev[0].record()
            s._use_current_stream()
            check(lib.nbody_order_compute(s._ctx, ptr(s.positions), n, ptr(order)), s._ctx)
            check(lib.nbody_order_gather(s._ctx, ptr(s.positions), ptr(s.positions), 0, n, 4), s._ctx)
            check(lib.nbody_order_gather(s._ctx, ptr(s.velocities), ptr(vel_all), lo, chunk, 4), s._ctx)
            check(lib.nbody_order_gather(s._ctx, ptr(order), ptr(order), 0, n, 2), s._ctx)
            ev[1].record()
            torch.cuda.synchronize()
            if it:
                refresh = min(refresh, ev[0].elapsed_time(ev[1]))
        print(f"    one rank's share of a layout refresh: {refresh:.3f} ms = {100 * refresh / (50 * best):.3f} % of 50 steps of {best:.3f} ms")
    s.close()
