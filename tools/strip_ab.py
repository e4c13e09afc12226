#!/usr/bin/env python3
"""Pair-once mode, one context: wall time per step, kernel times and partial-sum bytes with strips of 1 / 2 / 4 column splits
(nbody_set_strip_len) and summation parts, interleaved rounds in one process.  python tools/strip_ab.py [n] [steps] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    systems = []
    for strip, parts in ((1, 0), (4, 0), (4, 2), (4, 1)):
        s = nb.NBodySystem(n, split_len=nb.pair_once_split_len(n), body_order="morton")
        s.set_force_mode("pair_once")
        s.set_strip_len(strip)
        s.set_summation_parts(parts)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(2, 1e-3, 1e-3)
        systems.append((strip, parts, s))
    for rnd in range(rounds):
        for strip, parts, s in systems:
            t0 = time.perf_counter()
            s.step_n(steps, 1e-3, 1e-3)
            wall = (time.perf_counter() - t0) * 1e3 / steps
            s.timing(True)
            s.step_n(4, 1e-3, 1e-3)
            tm = s.read_timing()
            s.timing(False)
            print(f"round {rnd} n={n} strip {strip} parts {parts}: {wall:8.3f} ms/step  force {tm['force_ms'] / 4:8.3f} ({tm['force_launches'] // 4} launches) "
                  f"behind {tm['update_ms'] / 4:6.3f}  aux {tm.get('aux_ms', 0) / 4:6.3f}  partial sums {s.partial_sum_bytes() / 1e9:5.2f} GB", flush=True)


if __name__ == "__main__":
    main()
