"""Wall time per step against the kernels' own HIP-event times at small N: how much is launch overhead?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import n_body_problem_amd as nb

for n in (4096, 20225, 65536):
    pos, vel = nb.plummer(n, seed=7)
    for mode in ("one_sided", "pair_once"):
        s = nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0)
        s.set_force_mode(mode)
        s.setParticlesPosition(pos); s.setParticlesVelocity(vel)
        s.step_n(20, 1e-3, 1e-2)
        K = 200
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.step_n(K, 1e-3, 1e-2)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K
        s.timing(True)
        s.step_n(K, 1e-3, 1e-2)
        tm = s.read_timing()
        s.timing(False)
        print(f"N={n:6d} {mode:9s}: wall {wall*1e3:.4f} ms/step; force kernel {tm['force_ms']/K:.4f} ms, "
              f"update (+sum) {tm['update_ms']/K:.4f} ms; {n*n/wall:.3e} interactions/s")
        s.close()
