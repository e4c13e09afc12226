"""The body order computed ON THE DEVICE (csrc/nbody_order.hip, nbody_reorder / nbody_order_* of include/nbody.h) against the
host helper nbody_morton_order -- the same permutation bit for bit -- and the setters that rely on it.

The reference keeps its loader's order (kernel.cu:190-556) and its two setters are independent copies (kernel.cu:163-188);
the second half of this file holds the library to that: new positions keep every body's velocity, new velocities leave the
positions alone, whatever layout the device uses."""
import ctypes
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nb():
    import torch
    import n_body_problem_amd as nb
    assert torch.cuda.is_available(), "the gpu suite needs an MI355X"
    return nb


def device_order(nb, pos):
    """nbody_morton_order_device through the C ABI on a device copy of pos."""
    import torch
    from n_body_problem_amd import _lib
    lib = _lib.load()
    n = pos.shape[0]
    ctx = ctypes.c_void_p(None)
    assert lib.nbody_create(ctypes.byref(ctx), 0, n) == 0
    try:
        d = torch.from_numpy(np.ascontiguousarray(pos)).cuda()
        perm = torch.full((max(n, 1),), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        rc = lib.nbody_morton_order_device(ctx, ctypes.c_void_p(d.data_ptr() if n else 0), n, ctypes.c_void_p(perm.data_ptr()))
        assert rc == 0, lib.nbody_last_error(ctx)
        assert lib.nbody_sync(ctx) == 0
        return perm[:n].cpu().numpy()
    finally:
        lib.nbody_destroy(ctx)


def cases(nb):
    rng = np.random.default_rng(11)
    out = {}
    p, _ = nb.plummer(30000, seed=21)
    out["equal masses"] = p
    q = p.copy()
    q[1::3, 3] *= 2.0
    q[2::3, 3] *= 0.0                                  # three species, interleaved, one of them massless
    out["three species"] = q
    q = p.copy()
    q[:, 3] = rng.uniform(0.1, 1.0, len(q)).astype(np.float32)
    out["all masses distinct"] = q
    q = p.copy()
    q[:, 3] = rng.integers(1, 17, len(q)).astype(np.float32)          # exactly 16 species: still sorted by species
    out["sixteen species"] = q
    q = p.copy()
    q[:, 3] = rng.integers(1, 18, len(q)).astype(np.float32)          # 17: the curve alone
    out["seventeen species"] = q
    q = p[:5000].copy()
    q[100:400, :3] = q[100, :3]                        # coincident bodies: ties by index
    q[17, 0] = np.inf
    q[4, 2] = np.nan
    q[900, 1] = -np.inf
    out["ties and non-finite positions"] = q
    out["all at one point"] = np.zeros((777, 4), np.float32)
    q = np.zeros((300, 4), np.float32)
    q[:, 0] = np.linspace(-1.0, 1.0, 300, dtype=np.float32)           # a line: two axes without extent
    q[:, 3] = 1.0
    out["a line"] = q
    out["one body"] = np.array([[0.5, -2.0, 3.0, 1.0]], np.float32)
    out["no body"] = np.zeros((0, 4), np.float32)
    q, _ = nb.plummer(20000, seed=22)
    from n_body_problem_amd import initial_conditions as ic
    out["reference-style padding"] = ic.pad_reference_style(q, np.zeros_like(q))[0]   # 225 massless bodies at the origin
    return out


def test_device_order_is_the_host_order_bit_for_bit(nb):
    for name, pos in cases(nb).items():
        want = nb.morton_order(pos)
        got = device_order(nb, pos)
        assert np.array_equal(got, want), name


def test_device_order_at_the_headline_size_and_its_cost(nb):
    import torch
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    want = nb.morton_order(pos)
    assert np.array_equal(device_order(nb, pos), want)
    # the whole refresh (keys, sort, gathers of positions, velocities, order array) through NBodySystem.reorder()
    with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n), body_order="morton") as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        assert np.array_equal(s.order, want)
        s.reorder()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s.reorder()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        assert np.array_equal(s.order, want)           # nothing moved: the same curve
        p, v = s.download()
        assert np.array_equal(p, pos) and np.array_equal(v, vel)
    print(f"layout refresh at N = 2^20 on the device: {ms:.3f} ms")
    assert ms < 3.0                                     # VERDICT r02 item 2 (through the host: 130-300 ms)


@pytest.mark.parametrize("mode", ["pair_once", "one_sided"])
def test_setters_are_independent_copies_with_a_morton_layout(nb, mode):
    """a second setParticlesPosition used to re-sort the positions only, and every body silently carried another
    body's velocity and softening length."""
    n = 12000
    pos, vel = nb.plummer(n, seed=31)
    eps = np.random.default_rng(6).uniform(0.0, 0.02, n).astype(np.float32)
    pos2, _ = nb.plummer(n, seed=32)                   # other positions: another curve
    vel2 = (-vel).astype(np.float32)
    out = {}
    for order in ("morton", "given"):
        with nb.NBodySystem(n, split_len=512 if mode == "pair_once" else 0, body_order=order) as s:
            s.set_force_mode(mode)
            s.setParticlesVelocity(vel)                # velocities first: the order is still the identity
            s.setParticlesPosition(pos)
            s.set_particle_softening(eps)
            p, v = s.download()
            assert np.array_equal(p, pos) and np.array_equal(v, vel)
            s.step_n(2, 1e-3, 1e-3)
            _, v_mid = s.download()
            s.setParticlesPosition(pos2)               # the velocities (and eps) must stay with their bodies
            p, v = s.download()
            assert np.array_equal(p, pos2) and np.array_equal(v, v_mid)
            if order == "morton":
                assert np.array_equal(s.order, nb.morton_order(pos2))
                assert np.array_equal(s._eps_pp.cpu().numpy(), eps[s.order])
            s.step_n(2, 1e-3, 1e-3)
            p_mid, _ = s.download()
            s.setParticlesVelocity(vel2)               # and new velocities must not rewind the positions
            p, v = s.download()
            assert np.array_equal(p, p_mid) and np.array_equal(v, vel2)
            s.step_n(2, 1e-3, 1e-3)
            out[order] = s.download()
    scale_p, scale_v = np.abs(out["given"][0][:, :3]).max(), np.abs(out["given"][1][:, :3]).max()
    assert np.abs(out["morton"][0][:, :3] - out["given"][0][:, :3]).max() <= 1e-5 * scale_p
    assert np.abs(out["morton"][1][:, :3] - out["given"][1][:, :3]).max() <= 1e-5 * scale_v


@pytest.mark.parametrize("order", ["morton", "given"])
def test_multi_setters_are_independent_copies(nb, order):
    """MultiGpuSystem.setParticlesVelocity after some steps used to re-upload the positions last set from the
    host, rewinding them (and the other way round)."""
    from n_body_problem_amd.multi import MultiGpuSystem
    n = 20480
    pos, vel = nb.plummer(n, seed=33)
    pos2, _ = nb.plummer(n, seed=34)
    with MultiGpuSystem(n, devices=[0, 0], force_mode="pair_once", transport="peer_copy", body_order=order) as m:
        m.setParticlesPosition(pos)
        m.setParticlesVelocity(vel)
        p, v = m.download()
        assert np.array_equal(p, pos) and np.array_equal(v, vel)
        m.step_n(3, 1e-3, 1e-3)
        p_mid, v_mid = m.download()
        assert not np.array_equal(p_mid, pos)
        m.setParticlesVelocity(-vel)
        p, v = m.download()
        assert np.array_equal(p, p_mid) and np.array_equal(v, -vel)       # the positions were not rewound
        m.setParticlesPosition(pos2)
        p, v = m.download()
        assert np.array_equal(p, pos2) and np.array_equal(v, -vel)        # nor the velocities, nor dealt to other bodies
        if order == "morton":
            assert np.array_equal(m.order, nb.morton_order(pos2))
        m.step_n(2, 1e-3, 1e-3)
        got = m.download()
        assert m.replicas_identical()
    with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n)) as s:       # the same two steps on one context, by hand
        s.set_force_mode("pair_once")
        perm = nb.morton_order(pos2) if order == "morton" else np.arange(n)
        s.setParticlesPosition(pos2[perm])
        s.setParticlesVelocity(-vel[perm])
        s.step_n(2, 1e-3, 1e-3)
        want = s.download()
    assert np.array_equal(got[0][perm], want[0]) and np.array_equal(got[1][perm], want[1])


def test_reorder_through_the_c_abi_on_caller_owned_buffers(nb):
    """nbody_reorder as a C host calls it: positions, velocities, a softening array and an order array of its own."""
    import torch
    from n_body_problem_amd import _lib
    lib = _lib.load()
    n, n_real = 10240, 10000                            # a zero-mass padding tail stays where it is
    pos, vel = nb.plummer(n_real, seed=35)
    pp, vv = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    pp[:n_real], vv[:n_real] = pos, vel
    eps = np.random.default_rng(7).uniform(0.0, 0.01, n).astype(np.float32)
    ctx = ctypes.c_void_p(None)
    assert lib.nbody_create(ctypes.byref(ctx), 0, n) == 0
    try:
        dp, dv, de = torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda(), torch.from_numpy(eps).cuda()
        order = torch.empty(n_real, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        assert lib.nbody_order_identity(ctx, ptr(order), n_real) == 0
        assert lib.nbody_reorder(ctx, ptr(dp), ptr(dv), ptr(de), ptr(order), n_real) == 0
        assert lib.nbody_sync(ctx) == 0
        perm = nb.morton_order(pos)
        assert np.array_equal(order.cpu().numpy(), perm)
        assert np.array_equal(dp.cpu().numpy()[:n_real], pos[perm]) and np.array_equal(dv.cpu().numpy()[:n_real], vel[perm])
        assert np.array_equal(de.cpu().numpy()[:n_real], eps[:n_real][perm]) and np.array_equal(de.cpu().numpy()[n_real:], eps[n_real:])
        assert not dp.cpu().numpy()[n_real:].any()
        # a second refresh of the same state: the identity permutation, the order array unchanged
        assert lib.nbody_reorder(ctx, ptr(dp), ptr(dv), ptr(de), ptr(order), n_real) == 0
        assert lib.nbody_sync(ctx) == 0
        assert np.array_equal(order.cpu().numpy(), perm) and np.array_equal(dp.cpu().numpy()[:n_real], pos[perm])
        # bad arguments
        assert lib.nbody_reorder(ctx, ptr(dp), ptr(dv), None, None, n + 1) == _lib.NBODY_ERR_INVALID
        assert lib.nbody_reorder(ctx, None, ptr(dv), None, None, 5) == _lib.NBODY_ERR_INVALID
        assert lib.nbody_order_gather(ctx, ptr(dp), ptr(dp), 256, 256, 4) == _lib.NBODY_ERR_INVALID   # in place from row 0 only
    finally:
        lib.nbody_destroy(ctx)
    shard = ctypes.c_void_p(None)
    assert lib.nbody_create_shard(ctypes.byref(shard), 0, n, 0, n // 2, 256) == 0
    try:
        assert lib.nbody_reorder(shard, ptr(dp), ptr(dv), None, None, n) == _lib.NBODY_ERR_INVALID       # shards: nbody_multi_reorder
        assert lib.nbody_order_gather(shard, ptr(dp), ptr(dp), 0, n, 4) == _lib.NBODY_ERR_STATE          # no permutation yet
    finally:
        lib.nbody_destroy(shard)
