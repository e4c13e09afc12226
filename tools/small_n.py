#!/usr/bin/env python3
"""Step time at the reference's own problem size (galaxy_20K: N = 20000, the size its "1.6 ms" comment most
plausibly refers to) for a few split lengths and register blockings.  Development tool."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import n_body_problem_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pos, vel = nb.plummer(n, seed=7)
for split_len in (0, 256, 512, 1024, 2560):
    for rpl in (0, 1, 2, 4):
        try:
            s = nb.NBodySystem(n, split_len=split_len)
        except nb.NBodyError as e:
            print(split_len, "skip", e); continue
        s.set_rows_per_lane(rpl)
        s.setParticlesPosition(pos); s.setParticlesVelocity(vel)
        s.step_n(20, nb.TIME_TICK, nb.SOFTENING_VERSION3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 200
        s.step_n(K, nb.TIME_TICK, nb.SOFTENING_VERSION3)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print(f"N={n} split_len={s.split_len:5d} rpl={rpl}: {dt*1e3:.4f} ms/step  {n*n/dt:.3e} interactions/s")
        s.close()
