"""softening = 0 (the zero-distance guard in every loop) in both force modes against softening > 0:
python tools/eps0_modes.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [65536, 262144, 1048576]:
    pos, vel = nb.plummer(n, seed=7)
    for mode in ("one_sided", "pair_once"):
        for eps in (1e-3, 0.0):
            with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0) as s:
                s.set_force_mode(mode)
                s.setParticlesPosition(pos)
                s.setParticlesVelocity(vel)
                K = max(3, min(200, int(4e11 / (float(n) * n))))
                s.step_n(2, 1e-3, eps)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.step_n(K, 1e-3, eps)
                torch.cuda.synchronize()
                print(f"N={n:8d} {mode:9s} softening {eps:g}: {(time.perf_counter() - t0) / K * 1e3:9.3f} ms/step", flush=True)
