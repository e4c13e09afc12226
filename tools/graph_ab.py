#!/usr/bin/env python3
"""A/B: nbody_step_n with and without the HIP-graph replay of a step, wall time per step at small N."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import n_body_problem_amd as nb  # noqa: E402

for n in (256, 4096, 20225, 65536):
    pos, vel = nb.plummer(n, seed=7)
    for mode in ("one_sided", "pair_once"):
        row = []
        for replay in (0, 1):
            s = nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0)
            s.set_force_mode(mode)
            s.set_graph_replay(replay)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(50, 1e-3, 1e-2)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.step_n(400, 1e-3, 1e-2)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 400)
            row.append(best)
            s.close()
        print(f"N={n:6d} {mode:9s}: eager {row[0] * 1e6:8.1f} us/step   graph replay {row[1] * 1e6:8.1f} us/step", flush=True)
