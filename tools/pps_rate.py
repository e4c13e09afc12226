"""Force-kernel rate with per-particle softening on vs off (one GPU): python tools/pps_rate.py [n]"""
import sys
import numpy as np
import n_body_problem_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pos, vel = nb.plummer(n, seed=1)
eps = np.random.default_rng(1).uniform(0.0, 0.01, n).astype(np.float32)
with nb.NBodySystem(n) as s:
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    for label, e in (("global softening (asm kernel)", None), ("per-particle softening", eps)):
        s.set_particle_softening(e)
        s.step(1e-3, 1e-3)
        s.timing(True)
        s.step_n(3, 1e-3, 1e-3)
        s.sync()
        t = s.read_timing()
        s.timing(False)
        ms = t["force_ms"] / max(t["force_launches"], 1)
        print(f"{label}: {ms:.2f} ms/launch  {n * n / ms / 1e9:.3f}e12 interactions/s")
