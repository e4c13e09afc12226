#!/usr/bin/env python3
"""Development tool: scan small configurations of the library-owned exchange against one context (one GPU, peer copies)."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402
from n_body_problem_amd.multi import MultiGpuSystem  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    bad = 0
    for n, L, mode, world, exchange, integrator, pattern, eps, steps in itertools.product(
            (1316, 5000), (1024, 2048, 4096), ("pair_once", "one_sided"), (2, 4), ("allgather", "ring"), ("kick_drift", "kdk"),
            ("equal", "random"), (0.0, 1e-2), (1, 2)):
        pos = np.empty((n, 4), np.float32)
        pos[:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
        pos[:, 3] = 1.0 if pattern == "equal" else rng.uniform(0.0, 2.0, n).astype(np.float32)
        vel = (rng.normal(size=(n, 4)) * 0.1).astype(np.float32)
        with MultiGpuSystem(n, devices=[0] * world, force_mode=mode, integrator=integrator, exchange=exchange,
                            transport="peer_copy", split_len=L) as m:
            m.set_state(pos, vel)
            m.step_n(steps, 1e-3, eps)
            got = m.download()
            n_padded = m.n_padded
        pp, vv = np.zeros((n_padded, 4), np.float32), np.zeros((n_padded, 4), np.float32)
        pp[:n], vv[:n] = pos, vel
        with nb.NBodySystem(n_padded, split_len=L) as s:
            s.set_force_mode(mode)
            s.set_integrator(integrator)
            s.setParticlesPosition(pp)
            s.setParticlesVelocity(vv)
            s.step_n(steps, 1e-3, eps)
            want = s.download()
        ok = np.array_equal(got[0], want[0][:n]) and np.array_equal(got[1], want[1][:n])
        if not ok:
            bad += 1
            dv = np.abs(got[1][:, :3] - want[1][:n, :3])
            print(f"MISMATCH n={n} L={L} {mode} P={world} {exchange} {integrator} {pattern} eps={eps} steps={steps}: "
                  f"max dv {dv.max():.3e} rows differing {int((dv.max(1) > 0).sum())} first {int(np.argmax(dv.max(1) > 0))} "
                  f"nan {int(np.isnan(got[1]).sum())}/{int(np.isnan(want[1]).sum())}", flush=True)
    print("mismatches:", bad)


if __name__ == "__main__":
    main()
