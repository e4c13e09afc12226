#!/usr/bin/env python3
"""A/B on one GPU: the force pass at N = 2^20 with and without the equal-mass inner loop, both force modes; and the
pair-once mode on a body set with random masses (no split qualifies).  HIP-event times of the dominant kernel.
python tools/ab_equal_mass.py [steps=3] [n=1048576]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nb, n, pos, vel, mode, on, steps):
    with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0) as s:
        s.set_force_mode(mode)
        s.set_equal_mass_path(on)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.timing(True)
        s.step(1e-3, 1e-3)
        s.read_timing()
        for _ in range(steps):
            s.step(1e-3, 1e-3, sync=False)
        s.sync()
        t = s.read_timing()
    return t["force_ms"] / max(t["force_launches"], 1), t["update_ms"] / steps


def main():
    import n_body_problem_amd as nb
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    rnd = pos.copy()
    rnd[:, 3] *= (0.5 + np.random.default_rng(1).random(n)).astype(np.float32)
    out = {}
    for mode in ("pair_once", "one_sided"):
        for label, state, on in (("equal masses, short loop", pos, True), ("equal masses, general loop", pos, False),
                                 ("random masses", rnd, True)):
            f, u = run(nb, n, state, vel, mode, on, steps)
            pairs = n * n / 2 if mode == "pair_once" else n * n
            out[f"{mode}: {label}"] = {"force_ms": round(f, 3), "update_ms": round(u, 3),
                                       "frac_of_157.3_TF_at_20_flop_per_evaluation": round(20 * pairs / (f * 1e-3) / 157.3e12, 4)}
            print(f"{mode:10s} {label:28s} force {f:8.3f} ms  update {u:6.3f} ms  frac {out[f'{mode}: {label}']['frac_of_157.3_TF_at_20_flop_per_evaluation']:.4f}", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
