"""Step time of the two force modes by body count (what NBODY_PAIR_ONCE_MIN_BODIES / NBODY_FORCE_AUTO encode): wall time per
step over nbody_step_n, equal-mass Plummer sphere, the default split lengths of each mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import n_body_problem_amd as nb

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [24576, 32768, 40960, 49152, 57344, 65536, 98304, 131072, 196608, 262144]
for n in sizes:
    pos, vel = nb.plummer(n, seed=7)
    out = []
    for mode in ("one_sided", "pair_once"):
        with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n) if mode == "pair_once" else 0) as s:
            s.set_force_mode(mode)
            s.setParticlesPosition(pos); s.setParticlesVelocity(vel)
            K = max(20, min(2000, int(2e12 / (float(n) * n))))   # ~0.3 s per measurement: short bursts run at ramping clocks
            s.step_n(max(5, K // 4), 1e-3, 1e-2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.step_n(K, 1e-3, 1e-2)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / K * 1e3)
    frac = [20.0 * float(n) * n / (1.0 if m == 0 else 2.0) / (t * 1e-3) / 157.3e12 for m, t in enumerate(out)]
    print(f"N={n:7d}  one_sided {out[0]:8.4f} ms  pair_once {out[1]:8.4f} ms  ratio {out[1] / out[0]:.3f}   whole-step fraction of the "
          f"fp32 peak (20 flop x executed evaluations): {frac[0]:.3f} / {frac[1]:.3f}", flush=True)
