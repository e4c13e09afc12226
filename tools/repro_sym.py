#!/usr/bin/env python3
"""Development tool: pair-once against one-sided accelerations over split lengths, mass patterns and the equal-mass switch."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402


def accel(pos, eps, mode, L, on):
    n = pos.shape[0]
    with nb.NBodySystem(n, split_len=L) as s:
        s.set_force_mode(mode)
        s.set_equal_mass_path(on)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(pos))
        s.step(1.0, eps)
        return s.download()[1][:, :3].astype(np.float64)


def main():
    rng = np.random.default_rng(3)
    for n, L, pattern in itertools.product((3788, 700, 9000), (256, 512, 768, 1024, 1280, 2048, 4096), ("equal", "species", "random")):
        pos = np.empty((n, 4), np.float32)
        pos[:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
        if pattern == "equal":
            pos[:, 3] = 0.7
        elif pattern == "species":
            pos[:, 3] = 1.0
            pos[n // 3:, 3] = 0.25
            pos[2 * n // 3 + 17:, 3] = 3.0
        else:
            pos[:, 3] = rng.uniform(0.0, 2.0, n).astype(np.float32)
        one = accel(pos, 1e-3, "one_sided", L, False)
        for on in (True, False):
            pair = accel(pos, 1e-3, "pair_once", L, on)
            d = np.linalg.norm(pair - one) / np.linalg.norm(one)
            if not d < 1e-5:
                bad = np.linalg.norm(pair - one, axis=1) > 1e-4 * np.linalg.norm(one, axis=1).max()
                idx = np.nonzero(bad)[0]
                print(f"BAD n={n} L={L} {pattern} equal_mass_path={on}: rel diff {d:.3e}; rows off {idx.size}: {idx[:6]}..{idx[-3:]}", flush=True)
    print("done")


if __name__ == "__main__":
    main()
