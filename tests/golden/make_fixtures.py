"""Generates the committed golden fixtures from the CPU oracle.

The reference ships no golden vectors and cannot run here (oracle header: PARITY UNPINNED), so
these are the oracle's own outputs, kept so that (a) a later change to the oracle is caught and
(b) the HIP path is compared against fixed numbers as well as against a live oracle run.

    python tests/golden/make_fixtures.py        # rewrites tests/golden/f1_*.npz

Each file holds the inputs and, for K in STEPS, the state after K steps from the reference-order
fp32 path (``p32_K``, ``v32_K``) and from the fp64 truth (``p64_K``, ``v64_K``).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from n_body_problem_amd import initial_conditions as ic  # noqa: E402

STEPS = (1, 10, 100)
CASES = {
    # name: (generator, n, seed, dt, softening)
    "f1_cube64": ("cube", 64, 11, 1e-3, 1e-3),
    "f1_cube257": ("cube", 257, 12, 1e-3, 1e-3),
    "f1_plummer1024": ("plummer", 1024, ic.CONFIG_SEED[1], 1e-3, 1e-3),       # BASELINE.json configs[0]
    "f1_plummer1024_refconst": ("plummer", 1024, ic.CONFIG_SEED[1], 0.008, 1e-2),  # the reference's dt / eps
}


def generate(name):
    gen, n, seed, dt, eps = CASES[name]
    if gen == "cube":
        pos, vel = ic.uniform_cube(n, seed=seed, random_masses=True, speed=0.1)
    else:
        pos, vel = ic.plummer(n, seed=seed)
    out = {"pos0": pos, "vel0": vel, "dt": np.float32(dt), "softening": np.float32(eps),
           "steps": np.array(STEPS, dtype=np.int32)}
    for k in STEPS:
        p32, v32 = oracle.step_f32(pos, vel, dt, eps, nsteps=k, threads=1)
        p64, v64 = oracle.step_f64(pos, vel, dt, eps, nsteps=k, threads=1)
        out[f"p32_{k}"], out[f"v32_{k}"] = p32, v32
        out[f"p64_{k}"], out[f"v64_{k}"] = p64, v64
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **generate(name))
        print("wrote", name)
