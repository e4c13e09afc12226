#!/usr/bin/env python3
"""Probe (one GPU): does splitting the pair-once force pass into several launches cost tails, and do two streams hide them?
(a) one launch for all columns; (b) K column ranges, launches back to back on one stream; (c) the same launches dealt to
two streams.  Wall time of the force pass from torch events around it.  python tools/stream_overlap_probe.py [K=8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import n_body_problem_amd as nb
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    s = nb.NBodySystem(n, split_len=nb.pair_once_split_len(n))
    s.set_force_mode("pair_once")
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    chunk = n // k

    def run(kind):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(main_stream)
        if kind == "one":
            s.forces(0, n, 1e-3)
        else:
            for i in range(k):
                if kind == "two_streams" and i % 2:
                    side.wait_stream(main_stream) if i == 1 else None
                    with torch.cuda.stream(side):
                        s.forces(i * chunk, chunk, 1e-3)
                else:
                    s.forces(i * chunk, chunk, 1e-3)
            if kind == "two_streams":
                main_stream.wait_stream(side)
        b.record(main_stream)
        torch.cuda.synchronize()
        s.update(0.0)          # consumes the partial sums (dt = 0: the state does not move)
        s.sync()
        return a.elapsed_time(b)

    for kind in ("one", "one_stream", "two_streams", "one", "one_stream", "two_streams"):
        print(f"{kind:12s} K={k if kind != 'one' else 1:2d}  force pass {run(kind):8.3f} ms", flush=True)
    s.close()


if __name__ == "__main__":
    main()
