"""A few hundred steps of one context (nbody_create_auto's choice of mode and split length unless a mode is given) for
rocprofv3 --kernel-trace --stats: which launches a step consists of at this size and what each costs.
python tools/step_trace.py N [auto|one_sided|pair_once] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402

n = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "auto"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pos, vel = nb.plummer(n, seed=7)
s = nb.initialize(n, force_mode=mode)
s.setParticlesPosition(pos)
s.setParticlesVelocity(vel)
s.step_n(50, 1e-3, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
s.step_n(steps, 1e-3, 1e-3)
torch.cuda.synchronize()
print(f"N={n} {mode} (split_len {s.split_len}): {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step", flush=True)
