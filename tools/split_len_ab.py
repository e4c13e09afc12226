#!/usr/bin/env python3
"""Pair-once mode, one context: wall time per step for several split lengths, interleaved rounds in one process.
python tools/split_len_ab.py [n] [steps] [rounds] [len ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import n_body_problem_amd as nb  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    morton = "--morton" in sys.argv
    lens = [int(x) for x in sys.argv[4:] if x != "--morton"] or [1024, 2048]
    pos, vel = nb.plummer(n, seed=1)
    systems = []
    for L in lens:
        s = nb.NBodySystem(n, split_len=L, body_order="morton" if morton else "given")
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(2, 1e-3, 1e-3)
        s.sync()
        systems.append((L, s))
    for rnd in range(rounds):
        for L, s in systems:
            t0 = time.perf_counter()
            s.step_n(steps, 1e-3, 1e-3)
            s.sync()
            wall = (time.perf_counter() - t0) * 1e3 / steps
            print(f"round {rnd} n={n} split_len {L}: {wall:8.3f} ms/step  {float(n) * n / wall / 1e9:.3f}e12 interactions/s  "
                  f"partial sums {s.partial_sum_bytes() / 1e9:5.2f} GB", flush=True)


if __name__ == "__main__":
    main()
