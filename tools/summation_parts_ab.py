#!/usr/bin/env python3
"""Pair-once mode, one context: wall time per step, kernel times and partial-sum memory for 1, 2, 4 and 8 summation parts.
python tools/summation_parts_ab.py [n] [steps] [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    pos, vel = nb.plummer(n, seed=1)
    ref = None
    for rnd in range(rounds):
        for parts in (2, 8, 4, 1):
            with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if hasattr(nb, "pair_once_split_len") else 0) as s:
                s.set_force_mode("pair_once")
                s.set_summation_parts(parts)
                s.setParticlesPosition(pos)
                s.setParticlesVelocity(vel)
                s.step_n(2, 1e-3, 1e-3)
                s.sync()
                s.timing(True)
                s.read_timing()
                t0 = time.perf_counter()
                s.step_n(steps, 1e-3, 1e-3)
                s.sync()
                wall = (time.perf_counter() - t0) * 1e3 / steps
                tm = s.read_timing()
                got = s.download()
                if ref is None:
                    ref = got
                same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
                print(f"round {rnd} parts {parts}: {wall:8.3f} ms/step  force {tm['force_ms'] / steps:8.3f} ms in "
                      f"{tm['force_launches'] // steps} launches  behind {tm['update_ms'] / steps:6.3f} ms  beside "
                      f"{tm.get('aux_ms', 0.0) / steps:6.3f} ms  partial sums {s.partial_sum_bytes() / 1e9:6.2f} GB  "
                      f"free {torch.cuda.mem_get_info()[0] / 1e9:6.1f} GB  identical {same}", flush=True)
                assert same


if __name__ == "__main__":
    main()
