"""GPU: the sharded host path with the real HIP kernels.  A one-GPU box cannot run RCCL between ranks, so
the two-rank cases put both ranks on cuda:0 and let gloo carry the exchange: what is exercised is the
stream logic (own chunk beside the exchange, second-stream complement launch, ring hops) and the
bit-identity of P ranks with one context."""
import os
import socket

import numpy as np
import pytest

from _sharded_worker import run_rank_gpu

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def single(nb, n, steps):
    pos, vel = nb.plummer(n, seed=4321)
    with nb.NBodySystem(n) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, 1e-3, 1e-3)
        p, v = s.download()
        e = s.energy(1e-3)
    return p, v, e


def test_world_size_one_is_the_single_gpu_system():
    import n_body_problem_amd as nb
    from sharded_harness import ShardedNBodySystem
    n, steps = 20000, 3           # not a multiple of the split length: exercises the zero-mass padding
    p, v, e = single(nb, n, steps)
    pos, vel = nb.plummer(n, seed=4321)
    s = ShardedNBodySystem(n, device=0)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.step_n(steps, 1e-3, 1e-3)
    ps, vs = s.download()
    es = s.energy(1e-3)
    s.close()
    assert np.array_equal(ps, p) and np.array_equal(vs, v)
    assert np.allclose(es, e, rtol=1e-12)


@pytest.mark.parametrize("exchange", ["allgather", "ring"])
def test_two_ranks_on_one_gpu_reproduce_one_context_bit_for_bit(tmp_path, exchange):
    import torch.multiprocessing as mp
    import n_body_problem_amd as nb
    n, steps = 40000, 3
    p, v, e = single(nb, n, steps)
    out = str(tmp_path)
    mp.spawn(run_rank_gpu, args=(2, free_port(), exchange, n, steps, out), nprocs=2, join=True)
    for r in range(2):
        g = np.load(os.path.join(out, f"gpu_w2_{exchange}_r{r}.npz"))
        assert int(g["split_len"]) == nb.default_split_len(n)
        assert np.array_equal(g["p"], p) and np.array_equal(g["v"], v), (exchange, r)
        assert np.allclose(g["e"], e, rtol=1e-9)


def test_two_ranks_kdk_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    import n_body_problem_amd as nb
    n, steps = 40000, 3
    pos, vel = nb.plummer(n, seed=4321)
    with nb.NBodySystem(n) as s:
        s.set_integrator("kdk")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, 1e-3, 1e-3)
        p, v = s.download()
    out = str(tmp_path)
    mp.spawn(run_rank_gpu, args=(2, free_port(), "allgather", n, steps, out, "kdk"), nprocs=2, join=True)
    for r in range(2):
        g = np.load(os.path.join(out, f"gpu_w2_allgather_kdk_r{r}.npz"))
        assert np.array_equal(g["p"], p) and np.array_equal(g["v"], v), r


@pytest.mark.parametrize("world,exchange,integrator", [(2, "allgather", "kick_drift"), (2, "ring", "kick_drift"),
                                                       (4, "allgather", "kdk")])
def test_pair_once_ranks_on_one_gpu_reproduce_one_context_bit_for_bit(tmp_path, oracle_mod, world, exchange, integrator):
    """The pair-once mode sharded: 2 and 4 ranks (all on cuda:0, gloo carrying the positions and the column sums) end
    with the bits of one context on the same padded body set, and that state is the oracle's to rounding."""
    import torch.multiprocessing as mp
    import n_body_problem_amd as nb
    from sharded_harness import ShardedNBodySystem
    n, steps, split_len = 40000, 3, 512
    pos, vel = nb.plummer(n, seed=4321)
    s = ShardedNBodySystem(n, device=0, force_mode="pair_once", split_len=split_len, integrator=integrator)
    assert s.n_padded == 40960 and s.kernels.sym_groups() == (0, 8, 10)
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.step_n(steps, 1e-3, 1e-3)
    p, v = s.download()
    s.close()
    out = str(tmp_path)
    mp.spawn(run_rank_gpu, args=(world, free_port(), exchange, n, steps, out, integrator, "pair_once", split_len),
             nprocs=world, join=True)
    tag = (exchange if integrator == "kick_drift" else exchange + "_" + integrator) + "_pair_once"
    for r in range(world):
        g = np.load(os.path.join(out, f"gpu_w{world}_{tag}_r{r}.npz"))
        assert int(g["chunk"]) == 40960 // world
        assert np.array_equal(g["p"], p) and np.array_equal(g["v"], v), (exchange, r)
    step = oracle_mod.step_f32 if integrator == "kick_drift" else oracle_mod.step_kdk_f32
    pr, vr = step(pos, vel, 1e-3, 1e-3, nsteps=steps)
    from conftest import rel_state_error
    assert rel_state_error(p, pr) < 1e-6 and rel_state_error(v, vr) < 1e-6
