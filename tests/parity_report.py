#!/usr/bin/env python3
"""Measured parity on the reference's own inputs (tests/golden: galaxy_20K.bin, k17hp.snap, k17c.snap, stars_8192.dat), the reference's
way: padded to roundup(n, 256) + 1 (kernel.cu:260-278), dt = 0.008, VERSION 3's effective softening 1e-2 (kernel.cu:63-66,
665-692), K frames of the bracket kernel.cu:1225-1242.

For every input, force mode and body order: the HIP path's state against the fp64 truth (oracle.step_f64) and against the
oracle's literal restatement of VERSION 3 in the reference's fp32 order (oracle.step_v3), next to the error of that
restatement itself against the fp64 truth -- the number the GPU has to stay below to be "no worse than the reference order".
rel = max_i |x_i - ref_i|_inf / max_i |ref_i|_inf over the real bodies.  Run on an MI355X; the output is
committed under profiles/ and quoted in README.md / DESIGN.md next to the 1e-5 tolerance.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import n_body_problem_amd as nb
import oracle
from conftest import rel_state_error
from n_body_problem_amd import datasets as ds


def run(ppos, pvel, frames, mode, order):
    n = ppos.shape[0]
    with nb.NBodySystem(n, split_len=(256 if n < 8192 else nb.pair_once_split_len(n)) if mode == "pair_once" else 0,
                        body_order=order) as s:
        s.set_force_mode(mode)
        s.setParticlesPosition(ppos)
        s.setParticlesVelocity(pvel)
        s.step_n(frames, nb.TIME_TICK, nb.SOFTENING_VERSION3)
        return s.download()


def main():
    oracle.build()
    golden = os.path.join(ROOT, "tests", "golden")
    print(f"{'input':16s} {'frames':>6s} {'mode':10s} {'order':7s} | {'pos vs f64':>10s} {'vel vs f64':>10s} | "
          f"{'pos vs v3':>10s} {'vel vs v3':>10s} | oracle v3 vs f64: pos, vel")
    for name, frame_list in (("galaxy_20K.bin", (1, 10)), ("k17hp.snap", (1, 10)), ("k17c.snap", (3,)), ("stars_8192.dat", (1, 2))):
        pos, vel = ds.read_any(os.path.join(golden, name))
        ppos, pvel = nb.pad_reference_style(pos, vel)
        n = pos.shape[0]
        for frames in frame_list:
            p3, v3 = oracle.step_v3(ppos, pvel, nsteps=frames)
            p64, v64 = oracle.step_f64(ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, nsteps=frames)
            ref = (rel_state_error(p3[:n], p64[:n]), rel_state_error(v3[:n], v64[:n]))
            for mode in ("one_sided", "pair_once"):
                for order in ("given", "morton"):
                    p, v = run(ppos, pvel, frames, mode, order)
                    print(f"{name:16s} {frames:6d} {mode:10s} {order:7s} | {rel_state_error(p[:n], p64[:n]):10.3e} "
                          f"{rel_state_error(v[:n], v64[:n]):10.3e} | {rel_state_error(p[:n], p3[:n]):10.3e} "
                          f"{rel_state_error(v[:n], v3[:n]):10.3e} | {ref[0]:.3e}, {ref[1]:.3e}", flush=True)


if __name__ == "__main__":
    main()
