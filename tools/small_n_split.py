"""One-sided step at small N by split length (nbody_default_split_len's small-system rule, round 4): wall time per step over
200 steps of nbody_step_n and the kernels' own event times, interleaved rounds in one process.
python tools/small_n_split.py N[,N...] L[,L...]     (L = 0: the library's default)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import n_body_problem_amd as nb

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20225]
lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 0]
for n in sizes:
    pos, vel = nb.plummer(n, seed=7)
    systems = []
    for L in lens:
        s = nb.NBodySystem(n, split_len=L)
        s.setParticlesPosition(pos); s.setParticlesVelocity(vel)
        s.step_n(20, 8e-3, 1e-2)
        systems.append((L, s))
    for rnd in range(3):
        line = [f"N={n:6d} round {rnd}"]
        for L, s in systems:
            K = 300
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.step_n(K, 8e-3, 1e-2)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K
            s.timing(True)
            s.step_n(50, 8e-3, 1e-2)
            tm = s.read_timing()
            s.timing(False)
            line.append(f"split {s.split_len:4d}: {wall*1e3:7.4f} ms/step (force {tm['force_ms']/50:6.4f} update {tm['update_ms']/50:6.4f})")
        print("  ".join(line), flush=True)
    for _, s in systems:
        s.close()
