#!/usr/bin/env python3
"""A/B timing of force-kernel builds and register blockings in ONE process, interleaved rounds
(cdna_hip_programming.md section 5.4 rule 24).  Development tool.

    python tools/ab_force.py --n 1048576 --libs base=n_body_problem_amd/libnbody_amd.so,noslp=build/variants/noslp.so \
        --rpl 1,2,4,8 --rounds 5

Prints median / min milliseconds of one full force pass (all columns) and the implied interactions/s and
fraction of the 157.3 TFLOP/s fp32 vector peak at 20 flop per interaction.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(path):
    from n_body_problem_amd import _lib
    lib = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    for name, (res, args) in _lib._PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--libs", default="base=n_body_problem_amd/libnbody_amd.so")
    ap.add_argument("--rpl", default="1,2,4,8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--softening", type=float, default=1e-3)
    ap.add_argument("--split-len", type=int, default=0)
    ap.add_argument("--symmetric", action="store_true", help="time the experimental pair-once kernel (rpl is ignored)")
    ap.add_argument("--general-masses", action="store_true", help="equal-mass inner loops off: every split down the general path")
    ap.add_argument("--order", default="given", choices=["given", "morton", "radius", "hilbert"],
                    help="body order: as generated (random), along a Morton curve, or by distance from the centre (operand "
                         "toggling experiment: neighbours in index are neighbours in space)")
    args = ap.parse_args()

    import torch
    import n_body_problem_amd as nb
    pos, vel = nb.plummer(args.n, seed=nb.CONFIG_SEED[3])
    if args.order == "morton":
        q = np.clip(((pos[:, :3] + 4.0) / 8.0 * 1024).astype(np.int64), 0, 1023)
        key = np.zeros(args.n, np.int64)
        for b in range(10):
            for a in range(3):
                key |= ((q[:, a] >> b) & 1) << (3 * b + a)
        perm = np.argsort(key, kind="stable")
        pos, vel = np.ascontiguousarray(pos[perm]), np.ascontiguousarray(vel[perm])
        print("mean neighbour distance:", float(np.linalg.norm(np.diff(pos[:, :3], axis=0), axis=1).mean()))
    elif args.order == "hilbert":  # Skilling's transpose algorithm, 16 bits per axis
        bits = 16
        X = [np.clip(((pos[:, a] + 4.0) / 8.0 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1) for a in range(3)]
        M = 1 << (bits - 1)
        Q = M
        while Q > 1:
            P = Q - 1
            for i in range(3):
                hit = (X[i] & Q) != 0
                X[0] = np.where(hit, X[0] ^ P, X[0])
                t = np.where(hit, 0, (X[0] ^ X[i]) & P)
                X[0] ^= t
                X[i] ^= t
            Q >>= 1
        for i in range(1, 3):
            X[i] ^= X[i - 1]
        t = np.zeros_like(X[0])
        Q = M
        while Q > 1:
            t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
            Q >>= 1
        for i in range(3):
            X[i] ^= t
        key = np.zeros(args.n, np.int64)
        for b in range(bits - 1, -1, -1):
            for i in range(3):
                key = (key << 1) | ((X[i] >> b) & 1)
        perm = np.argsort(key, kind="stable")
        pos, vel = np.ascontiguousarray(pos[perm]), np.ascontiguousarray(vel[perm])
        print("mean neighbour distance:", float(np.linalg.norm(np.diff(pos[:, :3], axis=0), axis=1).mean()))
    elif args.order == "radius":
        perm = np.argsort(np.linalg.norm(pos[:, :3], axis=1), kind="stable")
        pos, vel = np.ascontiguousarray(pos[perm]), np.ascontiguousarray(vel[perm])
    dpos = torch.from_numpy(pos).cuda()
    dvel = torch.from_numpy(vel).cuda()
    torch.cuda.synchronize()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    variants = []
    for item in args.libs.split(","):
        name, path = item.split("=")
        lib = load(path)
        ctx = ctypes.c_void_p(None)
        rc = lib.nbody_create_shard(ctypes.byref(ctx), 0, args.n, 0, args.n, args.split_len)
        assert rc == 0, lib.nbody_last_error(None)
        lib.nbody_set_stream(ctx, stream)
        lib.nbody_timing_enable(ctx, 1)
        if args.symmetric:
            assert lib.nbody_set_force_mode(ctx, 1) == 0, lib.nbody_last_error(ctx)
        if args.general_masses:
            assert lib.nbody_set_equal_mass_path(ctx, 0) == 0
        for rpl in [int(x) for x in args.rpl.split(",")]:
            variants.append((f"{name}/rpl{rpl}", lib, ctx, rpl))

    def force_ms(lib, ctx, rpl):
        assert lib.nbody_set_rows_per_lane(ctx, rpl) == 0
        rc = lib.nbody_forces(ctx, ctypes.c_void_p(dpos.data_ptr()), 0, args.n, args.softening)
        assert rc == 0, lib.nbody_last_error(ctx)
        f_ms, u_ms = ctypes.c_double(0), ctypes.c_double(0)
        f_n, u_n = ctypes.c_int64(0), ctypes.c_int64(0)
        assert lib.nbody_timing_read(ctx, ctypes.byref(f_ms), ctypes.byref(f_n), ctypes.byref(u_ms), ctypes.byref(u_n)) == 0
        return f_ms.value

    for name, lib, ctx, rpl in variants:  # warm-up
        force_ms(lib, ctx, rpl)
    times = {v[0]: [] for v in variants}
    for _ in range(args.rounds):
        for name, lib, ctx, rpl in variants:
            times[name].append(force_ms(lib, ctx, rpl))
    inter = float(args.n) ** 2
    print(f"N={args.n} rounds={args.rounds} (one force pass = {inter:.3e} interactions)")
    for name in times:
        t = np.array(times[name])
        med, mn = np.median(t), t.min()
        print(f"{name:28s} median {med:9.3f} ms  min {mn:9.3f} ms  {inter / (med * 1e-3):.3e} int/s  "
              f"{20 * inter / (med * 1e-3) / 1e12:6.1f} TFLOP/s  {20 * inter / (med * 1e-3) / 157.3e12 * 100:5.1f}% of fp32 peak")


if __name__ == "__main__":
    main()
