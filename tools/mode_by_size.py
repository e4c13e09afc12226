#!/usr/bin/env python3
"""Step time of both force modes at one body count (best of three runs), the library's own split lengths and graph rule:
where does NBODY_FORCE_AUTO's choice stand?  (profiles/r04_pair_once_small_n.txt)
python tools/mode_by_size.py N [steps] [eps] [per-particle softening 0|1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
pps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pos, vel = nb.plummer(n, seed=7)
out = []
for mode in ("pair_once", "one_sided"):
    s = nb.initialize(n, force_mode=mode)
    if pps:
        s.set_particle_softening(np.full(n, 0.01, np.float32))
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.step_n(20, 1e-3, eps)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s.step_n(steps, 1e-3, eps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    out.append(f"{mode} (split {s.split_len}) {best:.4f}")
    s.close()
print(f"N={n} eps={eps} pps={pps}: " + "  ".join(out) + " ms/step", flush=True)
