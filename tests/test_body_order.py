"""nbody_morton_order (host helper of the C ABI) and the body_order option of the host layers.

The reference keeps whatever order its loader produced (kernel.cu:190-556) and order is not part of the physics; the
library may store the bodies along a Morton curve (fewer operand bits toggle between consecutive pair evaluations, the
power-limited clock rises) and must hand them back in the caller's order."""
import ctypes

import numpy as np
import pytest

import n_body_problem_amd as nb
from n_body_problem_amd import _lib


def test_morton_order_is_a_permutation_that_puts_neighbours_together():
    pos, _ = nb.plummer(20000, seed=5)
    perm = nb.morton_order(pos)
    assert perm.dtype == np.int64 and np.array_equal(np.sort(perm), np.arange(20000))
    step_given = np.linalg.norm(np.diff(pos[:, :3], axis=0), axis=1).mean()
    step_curve = np.linalg.norm(np.diff(pos[perm, :3], axis=0), axis=1).mean()
    assert step_curve < 0.3 * step_given                      # measured 0.12
    assert np.array_equal(perm, nb.morton_order(pos.copy()))  # a pure function of the bodies
    # the curve itself: 8 bodies at the corners of a cube come out in x-fastest, then y, then z order
    cube = np.array([[x, y, z, 1.0] for z in (0, 1) for y in (0, 1) for x in (0, 1)], np.float32)
    shuffled = np.random.default_rng(1).permutation(8)
    assert np.array_equal(cube[shuffled][nb.morton_order(cube[shuffled])], cube)


def test_morton_order_keeps_species_together_and_ignores_nothing():
    pos, _ = nb.plummer(6000, seed=6)
    pos[1::3, 3] *= 2.0                                       # two species interleaved in the caller's arrays
    pos[2::3, 3] *= 0.0                                       # and massless bodies
    m = pos[nb.morton_order(pos), 3]
    assert (np.diff(m) < 0).sum() == 0 and (np.diff(m) != 0).sum() == 2      # three runs, in order of mass
    rng = np.random.default_rng(2)
    pos[:, 3] = rng.uniform(0.1, 1.0, 6000).astype(np.float32)               # all masses distinct: the curve alone
    perm = nb.morton_order(pos)
    one = pos.copy()
    one[:, 3] = 1.0
    assert np.array_equal(perm, nb.morton_order(one))
    # equal positions: ties by index; non-finite positions last; empty and bad arguments
    same = np.zeros((5, 4), np.float32)
    assert np.array_equal(nb.morton_order(same), np.arange(5))
    bad = pos.copy()
    bad[17, 0] = np.inf
    bad[4, 2] = np.nan
    assert sorted(nb.morton_order(bad)[-2:].tolist()) == [4, 17]
    assert nb.morton_order(np.zeros((0, 4), np.float32)).shape == (0,)
    lib = _lib.load()
    assert lib.nbody_morton_order(None, 3, None) == _lib.NBODY_ERR_INVALID
    assert lib.nbody_morton_order(None, -1, None) == _lib.NBODY_ERR_INVALID


def test_body_order_argument_is_checked_without_a_device():
    with pytest.raises(ValueError):
        nb.NBodySystem(16, body_order="hilbert")
    from n_body_problem_amd.multi import MultiGpuSystem
    with pytest.raises(ValueError):
        MultiGpuSystem(16, body_order="hilbert")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pair_once", "one_sided"])
def test_morton_stored_system_is_the_sorted_input_bit_for_bit_and_the_same_physics(mode):
    """body_order="morton" = uploading the bodies in nbody_morton_order and undoing that at download: bit-identical to doing
    it by hand, and within the parity tolerance of the same bodies in the caller's order (another summation order)."""
    n, steps = 30000, 5
    pos, vel = nb.plummer(n, seed=8)
    pos[::2, 3] *= 3.0                                        # two species, interleaved
    vel[:, 3] = np.random.default_rng(3).uniform(0.0, 0.02, n).astype(np.float32)   # per-particle softening lengths
    perm = nb.morton_order(pos)
    out = {}
    for order in ("morton", "given", "by hand"):
        with nb.NBodySystem(n, split_len=1024 if mode == "pair_once" else 0, body_order="morton" if order == "morton" else "given") as s:
            s.set_force_mode(mode)
            p, v = (pos[perm], vel[perm]) if order == "by hand" else (pos, vel)
            s.setParticlesPosition(p)
            s.setParticlesVelocity(v)
            s.set_particle_softening(v[:, 3])
            s.step_n(steps, 1e-3, 1e-3)
            out[order] = s.download()
            if order == "morton":
                assert np.array_equal(s.order, perm)
                assert np.array_equal(s.positions.cpu().numpy()[:, 3], pos[perm, 3])     # the device order is the curve's
    assert np.array_equal(out["morton"][0][perm], out["by hand"][0]) and np.array_equal(out["morton"][1][perm], out["by hand"][1])
    scale_p, scale_v = np.abs(out["given"][0][:, :3]).max(), np.abs(out["given"][1][:, :3]).max()
    assert np.abs(out["morton"][0][:, :3] - out["given"][0][:, :3]).max() <= 1e-5 * scale_p
    assert np.abs(out["morton"][1][:, :3] - out["given"][1][:, :3]).max() <= 1e-5 * scale_v
    assert np.array_equal(out["morton"][0][:, 3], pos[:, 3]) and np.array_equal(out["morton"][1][:, 3], vel[:, 3])


@pytest.mark.gpu
def test_morton_stored_shards_equal_one_morton_stored_context():
    from n_body_problem_amd.multi import MultiGpuSystem
    n = 40000
    pos, vel = nb.plummer(n, seed=9)
    with MultiGpuSystem(n, devices=[0, 0, 0, 0], force_mode="pair_once", transport="peer_copy", body_order="morton") as m:
        m.set_state(pos, vel)
        m.step_n(3, 1e-3, 1e-3)
        got = m.download()
        assert m.replicas_identical()
        split_len, n_padded = m.split_len, m.n_padded
    perm = nb.morton_order(pos)
    pp, vv = np.zeros((n_padded, 4), np.float32), np.zeros((n_padded, 4), np.float32)
    pp[:n], vv[:n] = pos[perm], vel[perm]                     # the padding bodies follow the sorted real ones
    with nb.NBodySystem(n_padded, split_len=split_len) as s:
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pp)
        s.setParticlesVelocity(vv)
        s.step_n(3, 1e-3, 1e-3)
        want = s.download()
    assert np.array_equal(got[0][perm], want[0][:n]) and np.array_equal(got[1][perm], want[1][:n])


@pytest.mark.gpu
def test_headline_size_morton_stored_is_what_bench_runs(oracle_mod):
    """bench.py's default configuration: N = 2^20, pair-once mode, bodies stored along the Morton curve.  One force pass
    (zero velocities, dt = 1 leave the accelerations in the velocity buffer): sampled rows against the fp64 oracle, in the
    caller's order; Newton's third law over all pairs; and the whole field against the generator's order."""
    oracle = oracle_mod
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    acc = {}
    for order in ("morton", "given"):
        with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n), body_order=order) as s:
            s.set_force_mode("pair_once")
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(vel))
            s.step(1.0, 1e-3)
            p, v = s.download()
            assert np.array_equal(p[:, 3], pos[:, 3]) and np.all(v[:, 3] == 0)
            acc[order] = v[:, :3].astype(np.float64)
    for lo, hi in ((0, 128), (n // 2 - 64, n // 2 + 64), (n - 128, n)):
        a64 = oracle.accel_f64(pos, i0=lo, i1=hi, eps=1e-3)
        assert np.linalg.norm(acc["morton"][lo:hi] - a64) / np.linalg.norm(a64) < 1e-5
    mass = pos[:, 3].astype(np.float64)
    net = (mass[:, None] * acc["morton"]).sum(0)
    assert np.all(np.abs(net) < 1e-5 * (mass[:, None] * np.abs(acc["morton"])).sum(0))
    assert np.linalg.norm(acc["morton"] - acc["given"]) <= 1e-6 * np.linalg.norm(acc["given"])


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["kick_drift", "kdk"])
def test_reorder_refreshes_the_layout_and_keeps_the_physics(integrator):
    """nbody_multi_reorder / the reorder period: the state goes to the host and back in a new Morton order of the CURRENT
    positions (per-particle softening lengths follow their bodies).  With a period the run equals the same run with the
    reorder calls made by hand, bit for bit; against a run without refresh the state differs by rounding only; and two
    shards do what one does."""
    from n_body_problem_amd.multi import MultiGpuSystem
    n = 20480                                                   # 80 splits of 256: no padding in the pair-once geometry
    pos, vel = nb.plummer(n, seed=12)
    eps = np.random.default_rng(4).uniform(0.0, 0.02, n).astype(np.float32)
    out = {}
    for how in ("period", "by hand", "never", "two shards"):
        devices = [0, 0] if how == "two shards" else [0]
        with MultiGpuSystem(n, devices=devices, force_mode="pair_once", integrator=integrator, transport="peer_copy",
                            body_order="morton") as m:
            m.set_state(pos, vel)
            m.set_particle_softening(eps)
            first = m.order
            if how in ("period", "two shards"):
                m.set_reorder_period(4)
                m.step_n(10, 5e-3, 1e-3)                        # refreshes before steps 5 and 9
            elif how == "by hand":
                m.step_n(4, 5e-3, 1e-3)
                m.reorder()
                m.step_n(4, 5e-3, 1e-3)
                m.reorder()
                m.step_n(2, 5e-3, 1e-3)
            else:
                m.step_n(10, 5e-3, 1e-3)
            out[how] = m.download()
            if how == "period":
                assert not np.array_equal(m.order, first)       # the bodies have moved: another curve
                assert np.array_equal(np.sort(m.order), np.arange(n))
            if how == "never":
                assert np.array_equal(m.order, first)
    with MultiGpuSystem(n, devices=[0], force_mode="pair_once", integrator=integrator, transport="peer_copy") as probe:
        split_len, n_padded = probe.split_len, probe.n_padded
    assert n_padded == n
    if True:                                                    # the Python layer's period keeps the library's schedule
        with nb.NBodySystem(n, split_len=split_len, body_order="morton") as s:
            s.set_force_mode("pair_once")
            s.set_integrator(integrator)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.set_particle_softening(eps)
            s.set_reorder_period(4)
            s.step_n(10, 5e-3, 1e-3)
            out["python layer"] = s.download()
        assert np.array_equal(out["period"][0], out["python layer"][0]) and np.array_equal(out["period"][1], out["python layer"][1])
    for a in ("by hand", "two shards"):
        assert np.array_equal(out["period"][0], out[a][0]) and np.array_equal(out["period"][1], out[a][1]), a
    assert np.array_equal(out["period"][0][:, 3], pos[:, 3]) and np.array_equal(out["period"][1][:, 3], vel[:, 3])
    scale = np.abs(out["never"][0][:, :3]).max()
    assert 0 < np.abs(out["period"][0][:, :3] - out["never"][0][:, :3]).max() <= 1e-5 * scale


@pytest.mark.gpu
def test_single_context_reorder_in_the_python_layer():
    n = 20000
    pos, vel = nb.plummer(n, seed=13)
    eps = np.random.default_rng(5).uniform(0.0, 0.02, n).astype(np.float32)
    with nb.NBodySystem(n, body_order="morton") as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.set_particle_softening(eps)
        s.step_n(5, 5e-3, 1e-3)
        before = s.download()
        first = s.order.copy()
        s.reorder()
        assert not np.array_equal(s.order, first)
        after = s.download()
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])      # the same state, other slots
        assert np.array_equal(s._eps_pp.cpu().numpy(), eps[s.order])
        s.step_n(5, 5e-3, 1e-3)
        got = s.download()
    with nb.NBodySystem(n, body_order="given") as s:            # the same ten steps on the hand-sorted intermediate state
        perm = nb.morton_order(before[0])
        s.setParticlesPosition(before[0][perm])
        s.setParticlesVelocity(before[1][perm])
        s.set_particle_softening(eps[perm])
        s.step_n(5, 5e-3, 1e-3)
        want = s.download()
    assert np.array_equal(got[0][perm], want[0]) and np.array_equal(got[1][perm], want[1])
