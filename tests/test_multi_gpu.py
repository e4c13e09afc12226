"""GPU: the library-owned multi-GPU step (nbody_multi_*, csrc/nbody_multi.hip) with the real HIP kernels.

A one-GPU box cannot run RCCL between ranks (RCCL refuses two ranks on one device), so several shards on cuda:0 exchange
through NBODY_TRANSPORT_PEER_COPY -- the same streams, events, launch order and hazards, hipMemcpyPeerAsync in place of
the collective -- and must end with the bits of ONE context on the same padded system.  The RCCL leg itself runs with
the one rank the box has: communicator set-up, the in-place all-gather, the asynchronous-error poll."""
import numpy as np
import pytest

from conftest import rel_state_error

pytestmark = pytest.mark.gpu

DT, EPS = 1e-3, 1e-3


def one_context(nb, pos, vel, n_padded, split_len, steps, force_mode="one_sided", integrator="kick_drift", eps_pp=None):
    """The same padded body set on ONE context: what P shards must reproduce bit for bit."""
    return one_context_eps(nb, pos, vel, n_padded, split_len, steps, EPS, force_mode, integrator, eps_pp)


def one_context_eps(nb, pos, vel, n_padded, split_len, steps, eps, force_mode="one_sided", integrator="kick_drift", eps_pp=None):
    p = np.zeros((n_padded, 4), dtype=np.float32)
    v = np.zeros((n_padded, 4), dtype=np.float32)
    p[:pos.shape[0]], v[:vel.shape[0]] = pos, vel
    with nb.NBodySystem(n_padded, split_len=split_len) as s:
        s.set_force_mode(force_mode)
        s.set_integrator(integrator)
        if eps_pp is not None:
            e = np.zeros(n_padded, dtype=np.float32)
            e[:eps_pp.shape[0]] = eps_pp
            s.set_particle_softening(e)
        s.setParticlesPosition(p)
        s.setParticlesVelocity(v)
        s.step_n(steps, DT, eps)
        pp, vv = s.download()
        e = s.energy(eps)
        mom = s.momentum()
    n = pos.shape[0]
    return pp[:n], vv[:n], e, mom


@pytest.mark.parametrize("world,force_mode,exchange,integrator", [
    (2, "one_sided", "allgather", "kick_drift"), (2, "one_sided", "ring", "kick_drift"), (3, "one_sided", "ring", "kdk"),
    (2, "pair_once", "allgather", "kick_drift"), (2, "pair_once", "ring", "kdk"), (4, "pair_once", "allgather", "kdk"),
    (8, "pair_once", "ring", "kick_drift")])
def test_shards_on_one_gpu_reproduce_one_context_bit_for_bit(world, force_mode, exchange, integrator):
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 40000, 3                 # not a multiple of anything: exercises the zero-mass padding
    split = 512 if force_mode == "pair_once" else 0
    pos, vel = nb.plummer(n, seed=4321)
    with MultiGpuSystem(n, devices=[0] * world, force_mode=force_mode, integrator=integrator, exchange=exchange,
                        transport="peer_copy", split_len=split) as m:
        assert m.world_size == world and m.local_ranks == world and m.info()["rccl_ranks"] == 0
        m.set_state(pos, vel)
        m.step(DT, EPS)                 # one synchronous step, then the rest back to back (exchange in flight)
        m.step_n(steps - 1, DT, EPS)
        p, v = m.download()
        e, mom = m.energy(EPS), m.momentum()
        assert m.replicas_identical()
        n_padded, split_len = m.n_padded, m.split_len
    want_p, want_v, want_e, want_mom = one_context(nb, pos, vel, n_padded, split_len, steps, force_mode, integrator)
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)
    assert np.allclose(e, want_e, rtol=1e-9) and np.allclose(mom, want_mom, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("split_len,world", [(2048, 4), (4096, 2)])
def test_padding_bodies_stay_harmless_without_softening(split_len, world):
    """softening = 0, pair-once, kick-drift-kick, a body set much smaller than its padding: the zero-mass padding bodies
    start at one point and the pair-once tiles move them apart by an ulp (a lane's summation order depends on the lane);
    pairs that close must still contribute nothing -- 0 x inv^3 with an overflowed inv^3 used to be NaN for everybody."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 1316, 3
    rng = np.random.default_rng(12)
    pos = np.empty((n, 4), np.float32)
    pos[:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
    pos[:, 3] = 1.0
    vel = (rng.normal(size=(n, 4)) * 0.1).astype(np.float32)
    with MultiGpuSystem(n, devices=[0] * world, force_mode="pair_once", integrator="kdk", exchange="ring",
                        transport="peer_copy", split_len=split_len) as m:
        m.set_state(pos, vel)
        m.step_n(steps, DT, 0.0)
        p, v = m.download()
        e = m.energy(0.0)
        n_padded, L = m.n_padded, m.split_len
    assert n_padded == 8 * split_len and np.isfinite(p).all() and np.isfinite(v).all() and np.isfinite(e).all()
    want_p, want_v, _, _ = one_context_eps(nb, pos, vel, n_padded, L, steps, 0.0, "pair_once", "kdk")
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)


@pytest.mark.parametrize("n", [0, 1, 7, 300])
def test_empty_and_tiny_body_sets_through_the_multi_object(n, oracle_mod):
    """Edge sizes (the reference's loaders can return an empty vector, kernel.cu:195-199): nothing but padding on most
    ranks, every call still valid, the real bodies' result the oracle's."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    pos, vel = nb.uniform_cube(n, seed=3, random_masses=True, speed=0.2)
    for mode, world in (("one_sided", 3), ("pair_once", 8)):
        with MultiGpuSystem(n, devices=[0] * world, force_mode=mode, transport="peer_copy", exchange="ring") as m:
            assert m.n_padded % world == 0 and m.n_padded >= max(n, 1)
            m.set_state(pos, vel)
            m.step_n(2, DT, EPS)
            p, v = m.download()
            e, mom = m.energy(EPS), m.momentum()
            assert m.replicas_identical()
        assert p.shape == (n, 4) and v.shape == (n, 4) and np.isfinite(e).all() and np.isfinite(mom).all()
        if n:
            pr, vr = oracle_mod.step_f32(pos, vel, DT, EPS, nsteps=2)
            assert rel_state_error(p, pr) < 1e-6 and np.abs(v[:, :3] - vr[:, :3]).max() <= 1e-6 * max(np.abs(vr[:, :3]).max(), 1e-30) + 1e-9
            assert np.array_equal(p[:, 3], pos[:, 3]) and np.array_equal(v[:, 3], vel[:, 3])


def test_rccl_leg_with_the_one_rank_this_box_has(oracle_mod):
    """ncclCommInitAll / ncclCommInitRank, the in-place ncclAllGather (a no-op copy with one rank is never issued: world 1
    skips the exchange, so the communicator is exercised by the diagnostics' collectives) and the error poll."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem, unique_id
    n, steps = 8192, 2
    pos, vel = nb.plummer(n, seed=77)
    pr, vr = oracle_mod.step_f32(pos, vel, DT, EPS, nsteps=steps)
    want = None
    for make in (lambda: MultiGpuSystem(n, devices=[0], transport="rccl"),
                 lambda: MultiGpuSystem(n, devices=[0], transport="rccl", _rank=0, _world_size=1, _unique_id=unique_id())):
        with make() as m:
            assert m.info()["rccl_ranks"] == 1 and m.world_size == 1
            m.set_state(pos, vel)
            m.step_n(steps, DT, EPS)
            p, v = m.download()
            assert m.replicas_identical()
            e = m.energy(EPS)
        assert rel_state_error(p, pr) < 1e-6 and rel_state_error(v, vr) < 1e-6
        if want is None:
            want = (p, v, e)
        else:
            assert np.array_equal(p, want[0]) and np.array_equal(v, want[1]) and np.array_equal(e, want[2])


def test_from_torch_distributed_without_a_process_group_is_one_rank():
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import sharded_system
    from n_body_problem_amd.multi import MultiGpuSystem
    n = 5000
    pos, vel = nb.plummer(n, seed=5)
    s = sharded_system(n, device=0, force_mode="pair_once")
    assert isinstance(s, MultiGpuSystem) and s.world_size == 1
    s.setParticlesPosition(pos)
    s.setParticlesVelocity(vel)
    s.step_n(2, DT, EPS)
    p, v = s.download()
    n_padded, split_len = s.n_padded, s.split_len
    s.close()
    want_p, want_v, _, _ = one_context(nb, pos, vel, n_padded, split_len, 2, "pair_once")
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)


def test_particle_softening_and_timing_views_on_shards():
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 12000, 2
    pos, vel = nb.plummer(n, seed=9)
    eps_pp = (np.random.default_rng(3).random(n) * 0.02).astype(np.float32)
    with MultiGpuSystem(n, devices=[0, 0], transport="peer_copy") as m:
        m.set_state(pos, vel)
        m.set_particle_softening(eps_pp)
        for i in range(2):
            m.shard(i).timing(True)
        m.step_n(steps, DT, EPS)
        p, v = m.download()
        views = [m.positions_tensor(i).cpu().numpy() for i in range(2)]      # the mapped-pointer analogue, per local rank
        vviews = [m.velocities_tensor(i).cpu().numpy() for i in range(2)]
        tm = [m.shard(i).read_timing() for i in range(2)]
        info = m.kernels.device_info()
        n_padded, split_len = m.n_padded, m.split_len
    assert "gfx950" in info["name"]
    assert np.array_equal(views[0][:n], p) and np.array_equal(views[1][:n], p)                 # zero-copy replica views
    assert np.array_equal(np.concatenate(vviews)[:n], v)
    for t in tm:                        # own chunk + complement per step, one update per step
        assert t["force_launches"] == 2 * steps and t["update_launches"] == steps and t["force_ms"] > 0
    want_p, want_v, _, _ = one_context(nb, pos, vel, n_padded, split_len, steps, eps_pp=eps_pp)
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)


@pytest.mark.parametrize("world,exchange", [(2, "allgather"), (4, "ring")])
def test_particle_softening_pair_once_shards_through_the_eight_row_loop(world, exchange):
    """Per-particle softening in the pair-once mode with splits of 1024 bodies (the hand-scheduled S10 loop): P shards with the
    library-owned exchange equal one context on the padded body set, bit for bit, masses arbitrary."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 30000, 2
    pos, vel = nb.plummer(n, seed=10)
    rng = np.random.default_rng(5)
    pos[:, 3] *= rng.uniform(0.5, 2.0, n).astype(np.float32)
    eps_pp = (rng.random(n) * 0.02).astype(np.float32)
    with MultiGpuSystem(n, devices=[0] * world, force_mode="pair_once", exchange=exchange, transport="peer_copy",
                        split_len=1024) as m:
        m.set_state(pos, vel)
        m.set_particle_softening(eps_pp)
        m.step_n(steps, DT, EPS)
        p, v = m.download()
        n_padded = m.n_padded
    want_p, want_v, _, _ = one_context(nb, pos, vel, n_padded, 1024, steps, force_mode="pair_once", eps_pp=eps_pp)
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)


def test_a_step_that_outlasts_the_timeout_is_reported_not_waited_for():
    """Failure detection: a wait longer than the timeout returns NBODY_ERR_DEVICE with a message (a dead peer looks like
    this from the survivor's side) instead of hanging."""
    import n_body_problem_amd as nb
    from n_body_problem_amd import _lib
    from n_body_problem_amd.multi import MultiGpuSystem
    n = 1 << 18
    pos, vel = nb.plummer(n, seed=1)
    with MultiGpuSystem(n, devices=[0, 0], transport="peer_copy") as m:
        m.set_state(pos, vel)
        m.step(DT, EPS)                 # warm: allocations, kernel load
        m.set_timeout(1e-3)             # a step takes ~15 ms at this size
        with pytest.raises(_lib.NBodyError) as e:
            m.step_n(3, DT, EPS)
        assert e.value.status == _lib.NBODY_ERR_DEVICE and "timed out" in str(e.value)
        m.set_timeout(600.0)
        m.sync()                        # the device work itself was fine and drains


@pytest.mark.parametrize("force_mode", ["pair_once", "one_sided"])
def test_headline_size_two_shards_on_one_gpu_equal_one_context(oracle_mod, force_mode):
    """BASELINE configs[2]/[3] at the headline size: N = 2^20, the pair-once mode with 2048-body splits (what bench.py
    runs), as one context and as two shards with the library-owned exchange: the same bits; the accelerations of sampled
    rows agree with the fp64 oracle and the total force vanishes."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 1 << 20, 2
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    with MultiGpuSystem(n, devices=[0, 0], force_mode=force_mode, transport="peer_copy") as m:
        assert m.split_len == (2048 if force_mode == "pair_once" else 8192) and m.n_padded == n and m.chunk == n // 2
        m.set_state(pos, vel)
        m.step_n(steps, DT, EPS)
        p, v = m.download()
        assert m.replicas_identical()
        split_len = m.split_len
    with nb.NBodySystem(n, split_len=split_len) as s:
        s.set_force_mode(force_mode)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, DT, EPS)
        p2, v2 = s.download()
        assert np.array_equal(p, p2) and np.array_equal(v, v2)
        # size-independent properties of one force pass: zero velocities and dt = 1 leave a in the velocity buffer
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(vel))
        s.step(1.0, EPS)
        p1, v1 = s.download()
    acc = v1[:, :3].astype(np.float64)
    for lo, hi in ((0, 128), (n // 2 - 64, n // 2 + 64), (n - 128, n)):
        a64 = oracle_mod.accel_f64(pos, i0=lo, i1=hi, eps=EPS)
        assert np.linalg.norm(acc[lo:hi] - a64) / np.linalg.norm(a64) < 1e-5
    mass = pos[:, 3].astype(np.float64)
    net = (mass[:, None] * acc).sum(0)
    assert np.all(np.abs(net) < 1e-5 * (mass[:, None] * np.abs(acc)).sum(0))      # Newton's third law over 1.1e12 pairs
    assert np.array_equal(p1[:, 3], pos[:, 3]) and np.all(v1[:, 3] == 0)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_strips_shard_like_single_tiles(world):
    """Round 4: with strips (2048-body splits, four column splits per tile workgroup) the blocks of four are absolute and never
    straddle a summation group or a rank's column chunk, so 2, 4 and 8 shards with the library-owned exchange still end with the
    bits of one context -- N = 65 536 with 2048-body splits: 32 splits, one group = one strip length, the tightest case."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 65536, 3
    pos, vel = nb.plummer(n, seed=41)
    pos[: n // 5, 3] *= 3.0                      # two species: strips of one mass and mixed ones
    with nb.NBodySystem(n, split_len=2048) as s:
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, DT, EPS)
        want_p, want_v = s.download()
    for exchange in ("allgather", "ring"):
        with MultiGpuSystem(n, devices=[0] * world, force_mode="pair_once", exchange=exchange, transport="peer_copy",
                            split_len=2048) as m:
            m.set_state(pos, vel)
            m.step_n(steps, DT, EPS)
            p, v = m.download()
            assert m.replicas_identical()
        assert np.array_equal(p, want_p) and np.array_equal(v, want_v), (world, exchange)


@pytest.mark.parametrize("integrator", ["kick_drift", "kdk"])
def test_config5_shape_thousand_steps_one_context_and_two_shards(integrator):
    """BASELINE configs[4] in shape, at a size the GPU suite can afford: a long run (1000 steps) of a Plummer sphere in the
    pair-once mode with the bodies along the Morton curve, the layout refreshed every 50 steps on the device, energy checked
    at the end -- N = 2^17 instead of 2^22, as ONE rank and as TWO shards with the library-owned exchange: the same bits after
    1000 steps (a run's bits depend on the refresh period, not on the rank count), |dE/E0| bounded, momentum conserved."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 1 << 17, 1000
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[5])
    out = {}
    for world in (1, 2):
        with MultiGpuSystem(n, devices=[0] * world, force_mode="pair_once", integrator=integrator, transport="peer_copy",
                            body_order="morton") as m:
            assert m.split_len == 1024 and m.n_padded == n
            m.set_state(pos, vel)
            m.set_reorder_period(50)
            e0 = m.energy(EPS)
            m.step_n(steps, DT, EPS)
            e1, mom = m.energy(EPS), m.momentum()
            assert m.replicas_identical()
            out[world] = (*m.download(), e0, e1, mom)
    p1, v1, e0, e1, mom = out[1]
    p2, v2, e0b, e1b, _ = out[2]
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
    assert np.allclose(e0, e0b, rtol=1e-12) and np.allclose(e1, e1b, rtol=1e-12)
    drift = abs(e1[2] - e0[2]) / abs(e0[2])
    print(f"N = 2^17, 1000 steps, {integrator}, refresh every 50: |dE/E0| = {drift:.2e}, |p| = {np.abs(mom[:3]).max():.1e}")
    assert drift < (2e-5 if integrator == "kdk" else 2e-4)
    assert np.abs(mom[:3]).max() < 1e-5
    assert np.abs(p1[:, :3] - pos[:, :3]).max() > 0.1                       # a unit of time: the bodies have moved


def _one_rank_under_torch_nccl(rank, port, out_path):
    import torch
    import torch.distributed as dist
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    from n_body_problem_amd.multi import sharded_system
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 8192
        pos, vel = nb.plummer(n, seed=8)
        s = sharded_system(n, device=0, force_mode="pair_once", exchange="ring")     # backend nccl -> the library-owned exchange
        assert isinstance(s, MultiGpuSystem)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        dist.barrier()
        s.step_n(3, DT, EPS)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)                                                           # torch's communicator beside the library's
        p, v = s.download()
        # and bench.py's default layout: the bodies stored along the Morton curve, through the same factory
        m = sharded_system(n, device=0, force_mode="pair_once", body_order="morton")
        m.setParticlesPosition(pos)
        m.setParticlesVelocity(vel)
        m.step_n(3, DT, EPS)
        pm, vm = m.download()
        np.savez(out_path, p=p, v=v, e=s.energy(EPS), rccl_ranks=s.info()["rccl_ranks"], same=s.replicas_identical(),
                 n_padded=s.n_padded, split_len=s.split_len, t=float(t.item()), pm=pm, vm=vm, order=m.order)
        m.close()
        s.close()
    finally:
        dist.destroy_process_group()


def test_library_exchange_beside_torch_distributed_nccl(tmp_path):
    """What every rank of `torchrun bench.py --gpus N` does, with the one rank this box has: torch.distributed on the nccl
    backend (= RCCL) carries the RCCL id and the barriers, the library owns a second communicator in the same process
    (both bind to the RCCL build PyTorch ships), and the step runs inside the library."""
    import socket
    import torch.multiprocessing as mp
    import n_body_problem_amd as nb
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = str(tmp_path / "nccl_one_rank.npz")
    mp.spawn(_one_rank_under_torch_nccl, args=(port, out), nprocs=1, join=True)
    g = np.load(out)
    assert int(g["rccl_ranks"]) == 1 and bool(g["same"]) and float(g["t"]) == 1.0
    pos, vel = nb.plummer(8192, seed=8)
    want_p, want_v, want_e, _ = one_context(nb, pos, vel, int(g["n_padded"]), int(g["split_len"]), 3, "pair_once")
    assert np.array_equal(g["p"], want_p) and np.array_equal(g["v"], want_v) and np.allclose(g["e"], want_e, rtol=1e-9)
    perm = nb.morton_order(pos)
    assert np.array_equal(g["order"], perm)
    sorted_p, sorted_v, _, _ = one_context(nb, pos[perm], vel[perm], int(g["n_padded"]), int(g["split_len"]), 3, "pair_once")
    assert np.array_equal(g["pm"][perm], sorted_p) and np.array_equal(g["vm"][perm], sorted_v)


@pytest.mark.parametrize("exchange,force_mode", [("allgather", "pair_once"), ("ring", "one_sided")])
def test_per_rank_breakdown_of_a_step(exchange, force_mode):
    """nbody_multi_timing_*: where a rank's step goes besides its kernels -- what a one-shot 8-GPU run needs in its own
    output to be diagnosed.  Timing changes no bit."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps, world = 40960, 4, 2
    pos, vel = nb.plummer(n, seed=19)
    out = {}
    for timed in (True, False):
        with MultiGpuSystem(n, devices=[0] * world, force_mode=force_mode, exchange=exchange, transport="peer_copy",
                            body_order="morton") as m:
            m.set_state(pos, vel)
            m.step(DT, EPS)
            if timed:
                m.timing(True)
                m.read_timing(0), m.read_timing(1)
            m.step_n(steps, DT, EPS)
            m.reorder()
            if timed:
                m.sync()
                tms = [m.read_timing(i) for i in range(world)]
                again = m.read_timing(0)
            m.step_n(1, DT, EPS)
            out[timed] = m.download()
    assert np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][1], out[False][1])
    for t in tms:
        assert t["steps"] == steps and t["host_enqueue_ms"] > 0
        assert t["force_ms"] > 0 and t["update_ms"] > 0
        assert t["pos_exchanges"] == steps and t["pos_exchange_comm_ms"] > 0          # one exchange issued per step
        hops = (world - 1) if exchange == "ring" else 1
        # the first timed step follows a settled one (nothing in flight to wait for)
        assert t["pos_exchange_waits"] == (steps - 1) * hops and t["pos_exchange_wait_ms"] > 0
        assert t["column_sum_exchanges"] == (steps if force_mode == "pair_once" else 0)
        assert t["reorders"] == 1 and 0 < t["reorder_ms"] < 50
    assert again["steps"] == 0 and again["force_ms"] == 0 and again["pos_exchanges"] == 0   # a read resets the totals


def test_bench_leaves_a_json_error_line_when_an_exchange_times_out():
    """bench.py --exchange-timeout: a rank that waits longer than the timeout for an exchange prints ONE JSON line with
    "error" and exits non-zero -- evidence instead of a kill at the driver's limit.  Rehearsed with two
    peer-copy ranks on cuda:0 and a timeout shorter than a step."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer_copy", "--single-device",
           "--bodies", "262144", "--steps", "3", "--warmup", "1", "--exchange-timeout", "0.001", "--no-cpu-baseline",
           "--no-extra-legs", "--no-sanity"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    lines = [json.loads(x) for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None and "timed out" in lines[0]["error"]
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["exchange_timeout_s"] == 0.001
    # and with a sane timeout the same command measures, with the per-rank breakdown in its line
    cmd[cmd.index("--exchange-timeout") + 1] = "60"
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([x for x in res.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "peer_copy" in line["config"]["backend"]
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    for r in line["per_rank"]:
        assert r["force_ms"] > 0 and r["pos_exchange_wait_ms"] >= 0 and r["column_sum_exchange_ms"] > 0 and r["host_enqueue_ms"] > 0
