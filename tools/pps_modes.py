"""Per-particle softening (eps_ij^2 = eps^2 + eps_i^2 + eps_j^2) in both force modes against the plain step,
and the pair-once mode's hand-scheduled eight-row loop (S10) against the compiler-scheduled kernel (rows_per_lane 4):
python tools/pps_modes.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [65536, 262144, 1048576]
for n in sizes:
    pos, vel = nb.plummer(n, seed=7)
    eps = np.random.default_rng(1).uniform(0.0, 0.01, n).astype(np.float32)
    for mode, pps, rpl in (("one_sided", False, 0), ("one_sided", True, 0), ("pair_once", False, 0), ("pair_once", True, 4),
                           ("pair_once", True, 0)):
        with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0) as s:
            s.set_force_mode(mode)
            if rpl:
                s.set_rows_per_lane(rpl)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            if pps:
                s.set_particle_softening(eps)
            K = max(3, min(200, int(4e11 / (float(n) * n))))
            s.step_n(2, 1e-3, 1e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.step_n(K, 1e-3, 1e-3)
            torch.cuda.synchronize()
            kernel = "" if not (pps and mode == "pair_once") else " (compiler-scheduled kernel)" if rpl else " (hand-scheduled loop)"
            print(f"N={n:8d} {mode:9s} per-particle softening {str(pps):5s}: {(time.perf_counter() - t0) / K * 1e3:9.3f} ms/step{kernel}",
                  flush=True)
