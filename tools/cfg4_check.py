#!/usr/bin/env python3
"""BASELINE config 4 without 8 GPUs: N = 2^20 sharded over P ranks that all sit on cuda:0 (gloo carries the two
all-gathers), K steps, against the one-context run of the same padded system: the states must be identical bit for bit.
python tools/cfg4_check.py [P=4] [K=3] [force_mode=pair_once]"""
import os
import socket
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.multiprocessing as mp
    import n_body_problem_amd as nb
    from n_body_problem_amd.sharded import ShardedNBodySystem
    from _sharded_worker import run_rank_gpu
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    mode = sys.argv[3] if len(sys.argv) > 3 else "pair_once"
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=4321)                    # the seed run_rank_gpu uses
    t0 = time.perf_counter()
    s = ShardedNBodySystem(n, device=0, force_mode=mode)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.step_n(steps, 1e-3, 1e-3)
    p, v = s.download()
    s.close()
    print(f"one context: {steps} steps in {time.perf_counter() - t0:.1f} s", flush=True)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = tempfile.mkdtemp()
    t0 = time.perf_counter()
    mp.spawn(run_rank_gpu, args=(world, port, "allgather", n, steps, out, "kick_drift", mode, 0), nprocs=world, join=True)
    print(f"{world} ranks on one GPU: {time.perf_counter() - t0:.1f} s", flush=True)
    tag = "allgather" + ("_" + mode if mode != "one_sided" else "")
    for r in range(world):
        g = np.load(os.path.join(out, f"gpu_w{world}_{tag}_r{r}.npz"))
        same = np.array_equal(g["p"], p) and np.array_equal(g["v"], v)
        print(f"rank {r}: chunk {int(g['chunk'])}, split_len {int(g['split_len'])}, state identical to one context: {same}")
        assert same
    print(f"config 4 check passed: N = {n}, P = {world}, K = {steps}, force mode {mode}")


if __name__ == "__main__":
    main()
