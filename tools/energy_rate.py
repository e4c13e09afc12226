#!/usr/bin/env python3
"""Time of one nbody_energy pass (an O(N^2) potential evaluation) beside one force pass.  python tools/energy_rate.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import n_body_problem_amd as nb
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    with nb.NBodySystem(n) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        for eps in (1e-3, 0.0):
            s.energy(eps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e = s.energy(eps)
            torch.cuda.synchronize()
            print(f"N={n} eps={eps:g}: energy pass {1e3 * (time.perf_counter() - t0):8.2f} ms  E={e[2]:.9e}", flush=True)
        s.timing(True)
        s.step(1e-3, 1e-3)
        print("one-sided force pass", s.read_timing()["force_ms"], "ms")


if __name__ == "__main__":
    main()
