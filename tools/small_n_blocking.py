"""One-sided step at small N by kernel / register blocking (nbody_set_rows_per_lane: 1, 2, 4 = packed hand-allocated loop with
1024-row workgroups, 41 = the same loop with one wave per workgroup, 0 = what the library picks): wall time per step over
200 steps of nbody_step_n and the kernels' own event times.  Development tool behind pick_rows_per_lane (nbody_capi.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import n_body_problem_amd as nb

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192, 12288, 16384, 20225, 32768, 49152, 65536]
for n in sizes:
    pos, vel = nb.plummer(n, seed=7)
    line = [f"N={n:6d}"]
    for rpl in (0, 1, 2, 4, 41):
        with nb.NBodySystem(n) as s:
            s.set_rows_per_lane(rpl)
            s.setParticlesPosition(pos); s.setParticlesVelocity(vel)
            s.step_n(20, 1e-3, 1e-2)
            K = 200
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.step_n(K, 1e-3, 1e-2)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K
            s.timing(True)
            s.step_n(50, 1e-3, 1e-2)
            tm = s.read_timing()
        line.append(f"rpl {rpl:2d}: {wall*1e3:7.4f} ms (force {tm['force_ms']/50:6.4f} update {tm['update_ms']/50:6.4f})")
    print("  ".join(line), flush=True)
