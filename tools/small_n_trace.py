"""One-sided step at the reference's own size (galaxy_20K.bin padded to 20225, dt and softening of VERSION 3) -- run under
rocprofv3 --kernel-trace to see what the launches of a step cost and what lies between them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import n_body_problem_amd as nb
from n_body_problem_amd import datasets

pos, vel = datasets.read_tipsy(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "galaxy_20K.bin"))
ppos, pvel = nb.pad_reference_style(pos, vel)
mode = sys.argv[1] if len(sys.argv) > 1 else "one_sided"
s = nb.NBodySystem(ppos.shape[0], split_len=nb.pair_once_split_len(ppos.shape[0]) if mode == "pair_once" else 0)
s.set_force_mode(mode)
s.setParticlesPosition(ppos); s.setParticlesVelocity(pvel)
s.step_n(20, nb.TIME_TICK, nb.SOFTENING_VERSION3)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 200
s.step_n(K, nb.TIME_TICK, nb.SOFTENING_VERSION3)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / K * 1e3:.4f} ms/step")
