"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Stated tolerances (fp32 arithmetic, v_rsq_f32 <= 1 ulp, summation order differs from
the oracle's single ascending chain):
  * one-step accelerations vs the fp64 truth: relative L2 error <= 1e-5 (measured ~1e-7);
  * state after K steps vs the reference-order fp32 oracle: max|x - x_ref|_inf / max|x_ref|_inf <= 1e-5,
    same for v (BASELINE.json configs[1]: N = 65 536, K = 10);
  * GPU vs GPU (different register blocking, row shards, column ranges): BIT-EXACT.
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from conftest import rel_state_error

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def nb():
    import torch
    import n_body_problem_amd as nb
    assert torch.cuda.is_available(), "the gpu suite needs an MI355X"
    return nb


def run_gpu(nb, pos, vel, dt, eps, nsteps, mode="one_sided", **kw):
    if mode == "pair_once":
        kw.setdefault("split_len", 256 if pos.shape[0] < 8192 else nb.pair_once_split_len(pos.shape[0]))
    with nb.NBodySystem(pos.shape[0], **kw) as s:
        s.set_force_mode(mode)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(nsteps, dt, eps)
        return s.download()


def gpu_accel(nb, pos, eps, rpl=0):
    """Accelerations recovered from one step with dt=1 and v=0: v_new = a exactly (fp64 FMA, then rounded)."""
    vel = np.zeros_like(pos)
    with nb.NBodySystem(pos.shape[0]) as s:
        s.set_rows_per_lane(rpl)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step(1.0, eps)
        return s.download()[1][:, :3]


# ---- known answers -----------------------------------------------------------

def test_unit_pair_known_answer(nb):
    pos = np.array([[0, 0, 0, 1], [1, 0, 0, 1]], dtype=np.float32)
    for eps in (0.0, 1e-3, 1e-2, 0.5):
        a = gpu_accel(nb, pos, eps)
        want = (1.0 + eps * eps) ** -1.5
        assert a[0, 0] == pytest.approx(want, rel=1e-6) and a[1, 0] == pytest.approx(-want, rel=1e-6)
        assert np.all(a[:, 1:] == 0)


def test_coincident_bodies_and_zero_softening(nb):
    pos = np.array([[0.5, 0.5, 0.5, 1.0], [0.5, 0.5, 0.5, 2.0], [1.5, 0.5, 0.5, 4.0]], dtype=np.float32)
    for eps in (0.0, 1e-3):
        a = gpu_accel(nb, pos, eps)
        assert np.all(np.isfinite(a))
        assert a[0, 0] == pytest.approx(4.0 / (1 + eps * eps) ** 1.5, rel=1e-6)
        assert np.array_equal(a[0], a[1])


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_linearity_in_mass_and_translation(nb, mode):
    """Size-independent properties: accelerations are linear in the masses (a power-of-two factor: bit for bit, since it
    only shifts exponents), and a rigid translation changes them by rounding only."""
    n = 30000
    pos, _ = nb.plummer(n, seed=21)
    zero = np.zeros_like(pos)
    a1 = run_gpu(nb, pos, zero, 1.0, 1e-2, 1, mode)[1][:, :3]
    heavy = pos.copy()
    heavy[:, 3] *= 4.0
    a4 = run_gpu(nb, heavy, zero, 1.0, 1e-2, 1, mode)[1][:, :3]
    assert np.array_equal(a4, 4.0 * a1)
    moved = pos.copy()
    moved[:, :3] += np.array([0.25, -0.5, 0.125], np.float32)      # exactly representable shifts of O(1) coordinates
    am = run_gpu(nb, moved, zero, 1.0, 1e-2, 1, mode)[1][:, :3]
    assert np.linalg.norm(am - a1) / np.linalg.norm(a1) < 1e-5


def test_initialize_picks_the_force_mode_by_size(nb, oracle_mod):
    # round 4: the pair-once kernels are the faster ones at every size (NBODY_PAIR_ONCE_MIN_BODIES = 0)
    for n, want_len in ((700, 256), (5000, nb.pair_once_split_len(5000)), (65536, nb.pair_once_split_len(65536))):
        pos, vel = nb.plummer(n, seed=3)
        with nb.initialize(n, force_mode="auto") as s:
            assert s.split_len == want_len and s.force_mode == "pair_once"
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step(1e-3, 1e-3)
            p, v = s.download()
        pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=1)
        assert rel_state_error(p, pr) < TOL and rel_state_error(v, vr) < TOL


def test_empty_and_single_body(nb):
    for mode in ("one_sided", "pair_once"):
        p, v = run_gpu(nb, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), 1e-3, 1e-3, 2, mode)
        assert p.shape == (0, 4) and v.shape == (0, 4)
    p, v = run_gpu(nb, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), 1e-3, 1e-3, 2)
    assert p.shape == (0, 4) and v.shape == (0, 4)
    pos = np.array([[1, 2, 3, 5]], dtype=np.float32)
    vel = np.array([[1, 0, 0, 9]], dtype=np.float32)
    p, v = run_gpu(nb, pos, vel, 0.5, 1e-3, 2)
    assert np.array_equal(p, np.array([[2, 2, 3, 5]], np.float32)) and np.array_equal(v, vel)


# ---- accelerations and steps against the oracle ------------------------------------------------------

@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
@pytest.mark.parametrize("n", [2, 63, 255, 256, 257, 1000, 4097, 20000])
def test_one_step_matches_oracle_ragged_sizes(nb, oracle_mod, n, mode):
    pos, vel = nb.uniform_cube(n, seed=100 + n, random_masses=True, speed=0.2)
    a = run_gpu(nb, pos, np.zeros_like(vel), 1.0, 1e-3, 1, mode)[1][:, :3]   # dt = 1, v0 = 0: v = a
    a64 = oracle_mod.accel_f64(pos, eps=1e-3)
    assert np.linalg.norm(a - a64) / np.linalg.norm(a64) < TOL
    p, v = run_gpu(nb, pos, vel, 1e-3, 1e-3, 3, mode)
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=3)
    assert rel_state_error(p, pr) < TOL and rel_state_error(v, vr) < TOL
    assert np.array_equal(p[:, 3], pos[:, 3]) and np.array_equal(v[:, 3], vel[:, 3])  # mass / vel.w untouched


def test_gpu_is_no_worse_than_reference_order_fp32(nb, oracle_mod):
    pos, _ = nb.plummer(8192, seed=77)
    a64 = oracle_mod.accel_f64(pos, eps=1e-3)
    e_gpu = np.linalg.norm(gpu_accel(nb, pos, 1e-3) - a64) / np.linalg.norm(a64)
    e_ref = np.linalg.norm(oracle_mod.accel_f32(pos, eps=1e-3) - a64) / np.linalg.norm(a64)
    assert e_gpu < TOL and e_gpu < 4 * e_ref + 1e-7


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_config2_n65536_ten_steps(nb, oracle_mod, mode):
    """BASELINE.json configs[1]: N = 65 536 fp32, LDS tile 256, final state vs CPU within 1e-5 rel."""
    pos, vel = nb.plummer(65536, seed=nb.CONFIG_SEED[2])
    p, v = run_gpu(nb, pos, vel, 1e-3, 1e-3, 10, mode)
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=10)
    assert rel_state_error(p, pr) < TOL and rel_state_error(v, vr) < TOL


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_golden_fixtures(nb, golden_dir, mode):
    files = sorted(glob.glob(os.path.join(golden_dir, "f1_*.npz")))
    assert len(files) >= 4
    for f in files:
        g = np.load(f)
        for k in g["steps"]:
            p, v = run_gpu(nb, g["pos0"], g["vel0"], float(g["dt"]), float(g["softening"]), int(k), mode)
            tol = TOL if k <= 10 else 5e-5  # 100 steps of a chaotic system: rounding differences grow
            assert rel_state_error(p, g[f"p64_{k}"]) < tol and rel_state_error(v, g[f"v64_{k}"]) < tol, (f, k)
            assert rel_state_error(p, g[f"p32_{k}"]) < tol, (f, k)


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_reference_constants_and_padding(nb, oracle_mod, mode):
    """The reference's own configuration: dt = 0.008, VERSION 3 softening, roundup(n,256)+1 zero-mass padding."""
    pos, vel = nb.plummer(3000, seed=5)
    ppos, pvel = nb.pad_reference_style(pos, vel)
    p, v = run_gpu(nb, pos, vel, nb.TIME_TICK, nb.SOFTENING_VERSION3, 5, mode)
    pp, vp = run_gpu(nb, ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, 5, mode)
    n = pos.shape[0]
    # padding bodies sit at the origin with zero mass: real bodies' results are unchanged to rounding
    # (not bit-exact: the split boundaries depend on n_total)
    assert rel_state_error(pp[:n], p) < 1e-6 and rel_state_error(vp[:n], v) < 1e-6
    p3, v3 = oracle_mod.step_v3(ppos, pvel, nsteps=5)
    assert rel_state_error(pp[:n], p3[:n]) < TOL and rel_state_error(vp[:n], v3[:n]) < TOL


_GALAXY = {}


def galaxy_oracle(nb, oracle_mod, golden_dir):
    """The reference's own input and what the CPU paths make of it (computed once per session)."""
    if not _GALAXY:
        import os
        from n_body_problem_amd import datasets as ds
        pos, vel = ds.read_tipsy(os.path.join(golden_dir, "galaxy_20K.bin"))
        ppos, pvel = nb.pad_reference_style(pos, vel)
        _GALAXY.update(n=pos.shape[0], ppos=ppos, pvel=pvel,
                       v3_1=oracle_mod.step_v3(ppos, pvel, nsteps=1), v3_10=oracle_mod.step_v3(ppos, pvel, nsteps=10),
                       f64_10=oracle_mod.step_f64(ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, nsteps=10))
    return _GALAXY


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_reference_dataset_galaxy_20k_ten_steps(nb, oracle_mod, golden_dir, mode):
    """The reference's own workload, load_data(0) (kernel.cu:975-981): data/galaxy_20K.bin (three mass species, real
    close pairs), padded its way to 20 225 bodies (:260-278), dt = 0.008 and VERSION 3's effective softening (:63-66,
    665-692), ten frames of the bracket :1225-1242 -- against the oracle's literal restatement of VERSION 3.

    On THIS input fp32 summation order matters more than on the synthetic spheres: bodies beside a heavy halo particle
    carry partial sums of ~100 in an acceleration of ~10, the reference-order fp32 accelerations are 9e-6 (relative L2)
    off the fp64 truth, and after ten frames the oracle's reference-order velocities are 5.2e-5 off it (the reference's own
    float atomics, kernel.cu:758-773, pick yet another order on every run).  The HIP path sums in fixed-length splits and
    tiles and stays 14 x closer to the truth -- measured, both force modes, both body orders
    (profiles/r03_parity_reference_inputs.txt): positions 4.0e-7, velocities 3.7e-6 ... 3.9e-6.  So the stated tolerance holds
    where it means something: one frame within 1e-5 of the restatement; ten frames within 1e-5 of the fp64 TRUTH, no further
    from it than the reference order is, and within 1.15 x the restatement's own error of the restatement."""
    g = galaxy_oracle(nb, oracle_mod, golden_dir)
    n, ppos, pvel = g["n"], g["ppos"], g["pvel"]
    assert n == 20000 and ppos.shape[0] == 20225
    p1, v1 = run_gpu(nb, ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, 1, mode)
    assert rel_state_error(p1[:n], g["v3_1"][0][:n]) < TOL and rel_state_error(v1[:n], g["v3_1"][1][:n]) < TOL
    p, v = run_gpu(nb, ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, 10, mode)
    (p3, v3), (p64, v64) = g["v3_10"], g["f64_10"]
    assert rel_state_error(p[:n], p3[:n]) < TOL and rel_state_error(p[:n], p64[:n]) < TOL
    oracle_vs_truth = rel_state_error(v3[:n], v64[:n])
    gpu_vs_truth = rel_state_error(v[:n], v64[:n])
    print(f"galaxy_20K.bin, ten frames, {mode}: velocities vs fp64 truth: HIP {gpu_vs_truth:.3e}, reference-order fp32 oracle "
          f"{oracle_vs_truth:.3e}; HIP vs oracle {rel_state_error(v[:n], v3[:n]):.3e}")
    assert 1e-5 < oracle_vs_truth < 2e-4                  # the fp32 reference order itself: ~5e-5 here
    assert gpu_vs_truth < TOL                             # the stated tolerance, against the truth (measured 3.7e-6 ... 3.9e-6)
    assert gpu_vs_truth <= 1.0 * oracle_vs_truth          # no worse than the reference order (measured 14 x better)
    # against the restatement itself the distance is the restatement's own error (HIP sits 14 x closer to the truth): a number,
    # not the triangle inequality -- measured 5.6e-5 against 5.2e-5 (profiles/r03_parity_reference_inputs.txt), a ratio of 1.08
    assert rel_state_error(v[:n], v3[:n]) <= 1.15 * oracle_vs_truth
    assert np.array_equal(p[:, 3], ppos[:, 3])          # masses untouched
    assert np.array_equal(v[:, 3], pvel[:, 3])          # the eps column the reference loads and never reads: preserved


_DATASET_ORACLE = {}


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
@pytest.mark.parametrize("name,frames", [("k17hp.snap", 10), ("k17c.snap", 3), ("stars_8192.dat", 1)])
def test_reference_datasets_k17hp_and_stars(nb, oracle_mod, golden_dir, name, frames, mode):
    """The other inputs load_data serves (kernel.cu:996-1011), as committed data fixtures (tests/golden/README.md):
    k17hp.snap and k17c.snap (load_data(5) and (4): 10 002 and 32 770 equal-mass bodies, through the .snap parser -- the
    reference feeds them to its .dat parser; k17c for three frames, the CPU restatement of 32 770 bodies being
    what takes the time) and the first 8192 records of stars.dat (load_data(3): "z y x vz vy vx", every mass 1).  Padded the
    reference's way, its dt and VERSION 3 softening, against the oracle's restatement of VERSION 3 and the fp64 truth.
    stars.dat with unit masses is a violent collapse at dt = 0.008 (speeds of 200 after two frames; the fp32 restatement
    itself is 5e-5 off the truth after two frames and 3 % after five), so it is held to ONE frame; k17hp to ten.  Measured
    (profiles/r03_parity_reference_inputs.txt): k17hp ten frames 2.0e-7 / 3.2e-7 against truth / restatement; stars one frame
    <= 6.4e-7 / 3.0e-6."""
    import os
    from n_body_problem_amd import datasets as ds
    if name not in _DATASET_ORACLE:                              # the CPU paths once per session, not once per force mode
        pos, vel = ds.read_any(os.path.join(golden_dir, name))
        ppos, pvel = nb.pad_reference_style(pos, vel)
        _DATASET_ORACLE[name] = (pos.shape[0], ppos, pvel, oracle_mod.step_v3(ppos, pvel, nsteps=frames),
                                 oracle_mod.step_f64(ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, nsteps=frames))
    n, ppos, pvel, (p3, v3), (p64, v64) = _DATASET_ORACLE[name]
    assert n == {"k17hp.snap": 10002, "k17c.snap": 32770, "stars_8192.dat": 8192}[name]
    oracle_vs_truth = (rel_state_error(p3[:n], p64[:n]), rel_state_error(v3[:n], v64[:n]))
    for order in ("given", "morton"):
        p, v = run_gpu(nb, ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, frames, mode, body_order=order)
        gpu_vs_truth = (rel_state_error(p[:n], p64[:n]), rel_state_error(v[:n], v64[:n]))
        assert rel_state_error(p[:n], p3[:n]) < TOL and rel_state_error(v[:n], v3[:n]) < TOL
        assert gpu_vs_truth[0] < TOL and gpu_vs_truth[1] < TOL
        assert gpu_vs_truth[1] <= 1.0 * oracle_vs_truth[1] + 1e-7      # no worse than the reference order
        assert np.array_equal(p[:, 3], ppos[:, 3]) and np.array_equal(v[:, 3], pvel[:, 3])


_STARS = {}


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_reference_dataset_stars_whole_one_frame(nb, oracle_mod, golden_dir, mode):
    """load_data(3) at its REAL size (kernel.cu:996-1000): stars.dat whole, 43 802 unit-mass bodies read "z y x vz vy vx",
    padded the reference's way to 44 033, its dt and VERSION 3 softening, ONE frame of the bracket kernel.cu:1225-1242 (unit
    masses at dt = 0.008 are a violent collapse: after two frames the fp32 restatement itself is 5e-5 off the truth, so the
    state is held to one frame) -- the whole state against the oracle's restatement of VERSION 3 and against the fp64 truth,
    and the accelerations of that frame: every row against the fp64 oracle, and Newton's third law over all 9.7e8 pairs."""
    import os
    from n_body_problem_amd import datasets as ds
    if not _STARS:
        pos, vel = ds.read_any(os.path.join(golden_dir, "stars.dat"))
        ppos, pvel = nb.pad_reference_style(pos, vel)
        _STARS.update(n=pos.shape[0], ppos=ppos, pvel=pvel, v3=oracle_mod.step_v3(ppos, pvel, nsteps=1),
                      f64=oracle_mod.step_f64(ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, nsteps=1),
                      a64=oracle_mod.accel_f64(ppos, eps=nb.SOFTENING_VERSION3))
    n, ppos, pvel = _STARS["n"], _STARS["ppos"], _STARS["pvel"]
    assert n == 43802 and ppos.shape[0] == 44033 and np.all(ppos[:n, 3] == 1) and np.all(ppos[n:] == 0)
    (p3, v3), (p64, v64), a64 = _STARS["v3"], _STARS["f64"], _STARS["a64"]
    oracle_vs_truth = (rel_state_error(p3[:n], p64[:n]), rel_state_error(v3[:n], v64[:n]))
    for order in ("given", "morton"):
        p, v = run_gpu(nb, ppos, pvel, nb.TIME_TICK, nb.SOFTENING_VERSION3, 1, mode, body_order=order)
        gpu_vs_truth = (rel_state_error(p[:n], p64[:n]), rel_state_error(v[:n], v64[:n]))
        gpu_vs_oracle = (rel_state_error(p[:n], p3[:n]), rel_state_error(v[:n], v3[:n]))
        print(f"stars.dat whole, one frame, {mode}, {order}: positions / velocities vs fp64 truth: HIP {gpu_vs_truth[0]:.2e} / "
              f"{gpu_vs_truth[1]:.2e}, reference-order fp32 oracle {oracle_vs_truth[0]:.2e} / {oracle_vs_truth[1]:.2e}; HIP vs oracle "
              f"{gpu_vs_oracle[0]:.2e} / {gpu_vs_oracle[1]:.2e}")
        # 43 802 unit masses at eps = 1e-2: single accelerations of 1e4, and the restatement's one ascending fp32 chain over 44 033
        # terms is itself further from the truth than the tolerance; the HIP path (fixed-length splits) is held to the tolerance
        # against the TRUTH, to "no worse than the reference order", and to the restatement within the restatement's own error
        assert gpu_vs_truth[0] < TOL and gpu_vs_truth[1] < TOL
        assert gpu_vs_truth[0] <= oracle_vs_truth[0] + 1e-7 and gpu_vs_truth[1] <= oracle_vs_truth[1] + 1e-7
        assert gpu_vs_oracle[0] <= 1.15 * oracle_vs_truth[0] + 1e-7 and gpu_vs_oracle[1] <= 1.15 * oracle_vs_truth[1] + 1e-7
        assert np.array_equal(p[:, 3], ppos[:, 3]) and np.array_equal(v[:, 3], pvel[:, 3])
    kw = {"split_len": nb.pair_once_split_len(ppos.shape[0])} if mode == "pair_once" else {}
    with nb.NBodySystem(ppos.shape[0], **kw) as s:
        s.set_force_mode(mode)
        s.setParticlesPosition(ppos)
        s.setParticlesVelocity(np.zeros_like(pvel))
        s.step(1.0, nb.SOFTENING_VERSION3)               # v = 0, dt = 1: the velocities now hold the accelerations
        acc = s.download()[1][:, :3].astype(np.float64)
    assert np.linalg.norm(acc[:n] - a64[:n]) / np.linalg.norm(a64[:n]) < TOL
    worst = np.abs(acc[:n] - a64[:n]).max() / np.abs(a64[:n]).max()
    assert worst < TOL, worst
    m = ppos[:, 3].astype(np.float64)
    net = (m[:, None] * acc).sum(0)
    assert np.all(np.abs(net) < 1e-5 * (m[:, None] * np.abs(acc)).sum(0))


def test_bench_configuration_after_thirty_steps_against_the_oracle(nb, oracle_mod):
    """bench.py's exact configuration -- N = 2^20 Plummer sphere (configs[2]), the bodies stored along the Morton curve, the
    pair-once mode with 2048-body splits and the automatic summation parts, dt = softening = 1e-3 -- with the layout
    refreshed every 10 steps, AFTER 30 steps: the accelerations the library computes on the EVOLVED positions (velocities
    zeroed, one step of dt = 1) against the fp64 oracle on the same positions, 768 sampled rows, and Newton's third law.
    The first-step checks above cannot see a layout refresh or a summation slot going wrong later."""
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n), body_order="morton") as s:
        s.set_force_mode("pair_once")
        s.set_reorder_period(10)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(30, 1e-3, 1e-3)
        p30, v30 = s.download()
        assert np.array_equal(p30[:, 3], pos[:, 3]) and np.array_equal(v30[:, 3], vel[:, 3])
        assert not np.array_equal(s.order, nb.morton_order(pos))          # the layout has been refreshed on the way (before
                                                                           # steps 11 and 21: the curve of the positions then)
        s.setParticlesVelocity(np.zeros_like(vel))                         # an independent copy: the positions stay
        s.step(1.0, 1e-3)
        acc = s.download()[1][:, :3].astype(np.float64)
    moved = np.abs(p30[:, :3] - pos[:, :3]).max()
    assert 1e-3 < moved < 1.0                                              # 30 steps of dt = 1e-3 at speeds of order one
    for lo, hi in ((0, 256), (n // 2 - 128, n // 2 + 128), (n - 256, n)):
        a64 = oracle_mod.accel_f64(p30, i0=lo, i1=hi, eps=1e-3)
        assert np.linalg.norm(acc[lo:hi] - a64) / np.linalg.norm(a64) < TOL
    m = pos[:, 3].astype(np.float64)
    net = (m[:, None] * acc).sum(0)
    assert np.all(np.abs(net) < 1e-5 * (m[:, None] * np.abs(acc)).sum(0))


# ---- bit-exact invariances -----------------------------------------------------------------------

def test_register_blocking_is_bit_exact(nb):
    pos, vel = nb.plummer(10000, seed=31)
    ref = None
    for rpl in (1, 2, 4, 40, -4, 8):  # 4 = hand-allocated packed-fp32 loop, 40 = one row per instruction, -4 = compiled
        with nb.NBodySystem(pos.shape[0]) as s:
            s.set_rows_per_lane(rpl)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(2, 1e-3, 1e-3)
            out = s.download()
        if ref is None:
            ref = out
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]), rpl


def test_register_blocking_is_bit_exact_with_split_lengths_that_are_no_tile_multiple(nb, oracle_mod):
    """Round 4: at the reference's own size (20 225 bodies) the default split is 320 columns (64 splits x 80 row blocks = five
    waves per SIMD instead of 6.25): every split is one 256-column tile and a 64-column one, and the last is ragged.  Every
    kernel variant walks exactly the split's columns (the hand-allocated loops take the number of four-column iterations as an
    operand): the same bits from all of them, the oracle's accelerations, and other bits than 256-column splits give."""
    n = 20225
    pos, vel = nb.plummer(n, seed=33)
    pos[:, 3] *= np.where(np.arange(n) < 15000, np.float32(1.0), np.float32(3.0))     # two species: equal-mass and mixed splits
    assert nb.default_split_len(n) == 320
    ref = None
    for rpl in (0, 41, 1, 2, 4, 40, -4, 8):   # 0 picks the one-wave packed kernel (41) at this size
        with nb.NBodySystem(n) as s:
            assert s.split_len == 320
            s.set_rows_per_lane(rpl)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(2, 1e-3, 1e-3)
            out = s.download()
        if ref is None:
            ref = out
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]), rpl
    with nb.NBodySystem(n, split_len=256) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(2, 1e-3, 1e-3)
        other = s.download()
    assert not np.array_equal(other[1], ref[1]) and rel_state_error(other[1], ref[1]) < 1e-6
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=2)
    assert rel_state_error(ref[0], pr) < 1e-6 and rel_state_error(ref[1], vr) < 1e-6
    eps_pp = np.random.default_rng(3).uniform(0.0, 0.03, n).astype(np.float32)          # and under per-particle softening
    got = {}
    for rpl in (0, 1, 4):
        with nb.NBodySystem(n) as s:
            s.set_rows_per_lane(rpl)
            s.set_particle_softening(eps_pp)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(vel))
            s.step(1.0, 1e-3)
            got[rpl] = s.download()[1][:, :3]
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[4])
    want = oracle_mod.accel_f64_pps(pos, eps_pp, 1e-3)
    assert np.linalg.norm(got[0] - want) / np.linalg.norm(want) < 2e-6


def test_row_shards_and_column_ranges_are_bit_exact(nb):
    """P logical shards on one device, columns fed chunk by chunk in a rotated order: same bits as one context."""
    import torch
    n, P, steps = 8192, 4, 3
    pos, vel = nb.plummer(n, seed=32)
    split = 512                       # 16 splits; shard = 2048 rows = 4 splits
    with nb.NBodySystem(n, split_len=split) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, 1e-3, 1e-3)
        want_p, want_v = s.download()
    chunk = n // P
    shards = [nb.NBodySystem(n, row_lo=r * chunk, row_count=chunk, split_len=split) for r in range(P)]
    replica = torch.from_numpy(pos).cuda()
    for r, s in enumerate(shards):
        s.setParticlesVelocity(vel)
    for _ in range(steps):
        new = replica.clone()
        for r, s in enumerate(shards):
            for h in range(P):  # own chunk first, then the ring order a rank would receive them in
                c = (r - h) % P
                s.forces(c * chunk, chunk, 1e-3, positions=replica)
            s.update(1e-3, positions=new)   # writes rows of shard r only
            s.sync()
        # "all-gather": every shard's rows were written into `new` from the old replica
        upd = replica.clone()
        for r in range(P):
            upd[r * chunk:(r + 1) * chunk] = new[r * chunk:(r + 1) * chunk]
        replica = upd
    got_v = np.concatenate([s.velocities.cpu().numpy() for s in shards])
    assert np.array_equal(replica.cpu().numpy(), want_p) and np.array_equal(got_v, want_v)
    for s in shards:
        s.close()


def test_forces_complement_is_the_same_launch_set(nb):
    pos, vel = nb.plummer(6000, seed=33)
    outs = []
    for mode in ("whole", "own+complement"):
        with nb.NBodySystem(6000, split_len=512) as s:
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            if mode == "whole":
                s.forces(0, 6000, 1e-3)
            else:
                s.forces(1024, 2048, 1e-3)
                s.forces_complement(1024, 2048, 1e-3)
            s.update(1e-3)
            s.sync()
            outs.append(s.download())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_update_refuses_missing_column_ranges(nb):
    pos, vel = nb.plummer(2048, seed=3)
    with nb.NBodySystem(2048, split_len=256) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.forces(0, 1024, 1e-3)
        with pytest.raises(nb.NBodyError):
            s.update(1e-3)
        with pytest.raises(nb.NBodyError):
            s.forces(100, 256, 1e-3)       # not split-aligned
        s.forces(1024, 1024, 1e-3)
        s.update(1e-3)
        s.sync()


# ---- the call surface ------------------------------------------------------------------------------

def test_functional_step_and_separate_masses(nb, oracle_mod):
    import torch
    pos, vel = nb.uniform_cube(1500, seed=8, random_masses=True, speed=0.1)
    masses = pos[:, 3].copy()
    scrambled = pos.copy()
    scrambled[:, 3] = 123.0                      # mass must come from the masses argument
    dp, dv = torch.from_numpy(scrambled).cuda(), torch.from_numpy(vel).cuda()
    nb.step(dp, dv, torch.from_numpy(masses).cuda(), 1e-3, 1e-3)
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=1)
    assert rel_state_error(dp.cpu().numpy(), pr) < TOL and rel_state_error(dv.cpu().numpy(), vr) < TOL
    assert np.array_equal(dp.cpu().numpy()[:, 3], masses)
    dp2, dv2 = torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda()
    nb.step(dp2, dv2, None, 1e-3, 1e-3)          # masses=None: mass is positions[:,3] (the reference's layout)
    assert torch.equal(dp, dp2) and torch.equal(dv, dv2)


def test_c_abi_owned_buffers_without_torch_tensors(nb, oracle_mod):
    """The path a C/C++ host takes: create, set_positions/velocities (host arrays), step_n, download."""
    from n_body_problem_amd import _lib
    lib = _lib.load()
    pos, vel = nb.plummer(5000, seed=14)
    ctx = ctypes.c_void_p(None)
    assert lib.nbody_create(ctypes.byref(ctx), 0, 5000) == 0
    assert lib.nbody_step_n(ctx, 1, 1e-3, 1e-3) == -5            # buffers never set
    assert b"nbody_set_positions" in lib.nbody_last_error(ctx)
    assert lib.nbody_set_positions(ctx, pos.ctypes.data_as(ctypes.c_void_p)) == 0
    assert lib.nbody_set_velocities(ctx, vel.ctypes.data_as(ctypes.c_void_p)) == 0
    assert lib.nbody_positions_device(ctx) and lib.nbody_velocities_device(ctx)
    assert lib.nbody_step_n(ctx, 4, 1e-3, 1e-3) == 0
    p, v = np.empty_like(pos), np.empty_like(vel)
    assert lib.nbody_download(ctx, p.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)) == 0
    assert lib.nbody_destroy(ctx) == 0
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=4)
    assert rel_state_error(p, pr) < TOL and rel_state_error(v, vr) < TOL
    p2, v2 = run_gpu(nb, pos, vel, 1e-3, 1e-3, 4)
    assert np.array_equal(p, p2) and np.array_equal(v, v2)


def test_bad_arguments(nb):
    with nb.NBodySystem(512) as s:
        with pytest.raises(nb.NBodyError):
            s.step(1e-3, -1.0)
        with pytest.raises(nb.NBodyError):
            s.step(float("nan"), 1e-3)
        with pytest.raises(ValueError):
            s.setParticlesPosition(np.zeros((5, 4), np.float32))


    with nb.NBodySystem(512) as s:                                   # 0 < eps < 1e-9: eps^-3 x mass would overflow fp32
        with pytest.raises(nb.NBodyError) as e:
            s.step(1e-3, 1e-12)
        assert "softening" in str(e.value)
        s.step(1e-3, 1e-9)                                            # the smallest accepted length
        s.step(1e-3, 0.0)                                             # and exactly 0
    # pair-once shards own whole splits: rows that end in the middle of one are refused (the diagonal tile would write
    # past the context's rows)
    with nb.NBodySystem(200, row_lo=0, row_count=100, split_len=256) as s:
        with pytest.raises(nb.NBodyError) as e:
            s.set_force_mode("pair_once")
        assert "split boundary" in str(e.value)
    with nb.NBodySystem(4096, row_lo=0, row_count=1000, split_len=256) as s:
        with pytest.raises(nb.NBodyError):
            s.set_force_mode("pair_once")
    with nb.NBodySystem(200, row_lo=0, row_count=200, split_len=256) as s:   # ending at n_total is fine
        s.set_force_mode("pair_once")


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
@pytest.mark.parametrize("heavy", [1.0e3, 1.0e10])
def test_zero_softening_is_safe_for_any_finite_mass(nb, oracle_mod, mode, heavy):
    """softening = 0 with heavy bodies: the self pair and coincident bodies contribute exactly 0 (a clamp of r^2 instead
    would give m x 1e36 = inf and 0 x inf = NaN in every sum).  Every kernel variant against the oracle, whose pair
    function returns 0 at zero distance (oracle/nbody_oracle.c, pair_f32)."""
    n = 3000
    pos, vel = nb.uniform_cube(n, seed=21, random_masses=True)
    pos[:, 3] *= heavy * n                             # masses in [0.5, 1.5] x heavy
    pos[7] = pos[8]                                    # coincident bodies, same split
    pos[9, :3] = pos[2000, :3]                         # coincident bodies, different splits
    pos[11, 3] = np.float32(3.0e38)                    # close to FLT_MAX, but far from everybody
    pos[11, :3] = (50.0, 60.0, 70.0)
    want = oracle_mod.accel_f64(pos, eps=0.0)
    assert np.all(np.isfinite(want))
    got = {}
    for rpl in ((0, 1, 2, 8, 40, -4) if mode == "one_sided" else (0,)):
        with nb.NBodySystem(n, split_len=256 if mode == "pair_once" else 0) as s:
            s.set_force_mode(mode)
            s.set_rows_per_lane(rpl)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(vel))
            s.step(1.0, 0.0)
            got[rpl] = s.download()[1][:, :3]
            e = s.energy(0.0)
        assert np.all(np.isfinite(got[rpl])) and np.all(np.isfinite(e))
        err = np.linalg.norm(got[rpl] - want, axis=1) / np.linalg.norm(want, axis=1)
        assert err.max() < 1e-4 and np.sqrt((err ** 2).mean()) < TOL
        assert np.array_equal(got[rpl], got[0])        # every register blocking: the same bits
    a = got[0]
    assert np.array_equal(a[7], a[8])                  # coincident bodies feel the same force, none from each other


@pytest.mark.parametrize("mode", ["one_sided", "pair_once"])
def test_equal_mass_splits_take_a_shorter_inner_loop_with_the_same_answer(nb, oracle_mod, mode):
    """Splits whose bodies share one mass leave the mass out of the inner loop (nbody_set_equal_mass_path): against the
    general path the result moves by rounding only, both agree with the fp64 truth, and a body set with a few species
    in index order (some tiles qualify, some do not -- galaxy_20K.bin's shape) and a ragged tail is handled tile by tile."""
    n = 9000
    pos, vel = nb.plummer(n, seed=77)                           # equal masses: every full split qualifies
    species = pos.copy()
    species[:2048, 3] *= 17.0                                   # three species in index order, boundaries on and off
    species[2048:5000, 3] *= 0.25                               # the 256-body split grid
    # split lengths: whole passes of the pair-once tile (256, 1024) and ones that leave a partial last pass (768 with two
    # waves, 1280 with four), whose dummy rows must stay out of the column sums
    for state, split_len in ((pos, 256), (species, 256), (pos, 768), (species, 1280), (pos, 1024)):
        if mode == "one_sided" and split_len != 256:
            continue
        acc = {}
        for on in (True, False):
            with nb.NBodySystem(n, split_len=split_len if mode == "pair_once" else 0) as s:
                s.set_force_mode(mode)
                s.set_equal_mass_path(on)
                s.setParticlesPosition(state)
                s.setParticlesVelocity(np.zeros_like(vel))
                s.step(1.0, 1e-2)
                acc[on] = s.download()[1][:, :3].astype(np.float64)
        a64 = oracle_mod.accel_f64(state, eps=1e-2)
        scale = np.linalg.norm(a64)
        assert np.linalg.norm(acc[True] - a64) / scale < TOL and np.linalg.norm(acc[False] - a64) / scale < TOL
        moved = np.linalg.norm(acc[True] - acc[False]) / scale
        assert moved < 1e-6                                                      # a different rounding, nothing more
        assert (moved > 0) == (mode == "one_sided" or split_len in (256, 1024))  # ... where the shorter loop applies at all
    # the register blockings of the one-sided kernel stay bit-identical on the short path too
    if mode == "one_sided":
        ref = None
        for rpl in (0, 1, 2, 4, 8, 40, 41, -4):
            with nb.NBodySystem(n) as s:
                s.set_rows_per_lane(rpl)
                s.setParticlesPosition(pos)
                s.setParticlesVelocity(vel)
                s.step_n(2, 1e-3, 1e-3)
                got = s.download()
            ref = ref or got
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), rpl


@pytest.mark.parametrize("mode,integrator", [("one_sided", "kick_drift"), ("pair_once", "kick_drift"), ("pair_once", "kdk"),
                                             ("one_sided", "kdk")])
def test_graph_replay_of_a_step_gives_the_same_bits(nb, mode, integrator):
    """nbody_step_n on small systems replays ONE captured step as a HIP graph: the same kernels in the same order, so the
    state must equal the eagerly launched loop bit for bit -- also when the masses change between two calls (the
    equal-mass flags are recomputed inside the graph) and when the loop is called again (the graph is reused)."""
    n = 6000
    pos, vel = nb.plummer(n, seed=3)
    out = {}
    for replay in (1, 0):
        with nb.NBodySystem(n, split_len=256 if mode == "pair_once" else 0) as s:
            s.set_force_mode(mode)
            s.set_integrator(integrator)
            s.set_graph_replay(replay)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(7, 1e-3, 1e-2)
            s.step_n(5, 1e-3, 1e-2)                       # the cached graph again
            s.positions[: n // 2, 3] *= 3.0               # half the bodies three times as heavy: other inner loops
            s.invalidate_forces()
            s.step_n(4, 1e-3, 1e-2)
            s.step_n(3, 2e-3, 1e-2)                       # another dt: a new graph
            out[replay] = s.download()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])


@pytest.mark.parametrize("integrator", ["kick_drift", "kdk"])
def test_summation_parts_change_no_bit(nb, integrator):
    """Pair-once mode, one context, a system large enough for several tile launches: the row groups go in 1, 2, 4 or 8
    launches, each part's sums formed on the auxiliary stream beside the next part's tiles; with 4 and 8 parts the partial
    sums live in two slots used in turn (a fraction of the memory).  The association is by groups in every case, so the state
    must be the same bit for bit (also against two shards, which always take one part)."""
    from n_body_problem_amd.multi import MultiGpuSystem
    n, steps = 1 << 18, 3                       # 256 splits of 1024: 32 640 tiles
    pos, vel = nb.plummer(n, seed=19)
    pos[: n // 3, 3] *= 2.0                     # two mass species: both inner loops run
    out, held = {}, {}
    for parts in (1, 2, 4, 8, 0):
        with nb.NBodySystem(n, split_len=1024) as s:
            s.set_force_mode("pair_once")
            s.set_integrator(integrator)
            s.set_summation_parts(parts)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(steps, 1e-3, 1e-3)
            out[parts] = s.download()
            held[parts] = s.partial_sum_bytes()
    for parts in (2, 4, 8, 0):
        assert np.array_equal(out[parts][0], out[1][0]) and np.array_equal(out[parts][1], out[1][1]), parts
    whole = 12 * n * n // 1024                  # n^2 / split_len entries of 12 bytes
    assert abs(held[1] - whole) < whole // 100 and abs(held[2] - whole) < whole // 100
    # 4 parts = 3 + 3 + 1 + 1 of the 8 groups in two slots of three groups; 8 parts = two slots of one group (+ the diagonal tiles' slot)
    assert held[4] <= whole * 3 // 4 * 1.01 and held[8] <= whole // 4 * 1.01, held
    with MultiGpuSystem(n, devices=[0, 0], force_mode="pair_once", integrator=integrator, transport="peer_copy") as m:
        m.set_state(pos, vel)
        m.step_n(steps, 1e-3, 1e-3)
        p, v = m.download()
    assert np.array_equal(p, out[1][0]) and np.array_equal(v, out[1][1])


@pytest.mark.parametrize("n", [65536, 131072])
def test_strips_keep_the_rows_sums_in_registers_and_the_results(nb, oracle_mod, n):
    """Round 4, strips: with 2048-body splits and a split count that is a multiple of 32 a tile workgroup takes four consecutive
    column splits of its row split and keeps the rows' sums in registers across them -- a quarter of the row-side partial sums.
    Against the fp64 oracle (sampled rows), against single tiles (nbody_set_strip_len(1): equal to rounding, not bit for bit:
    one chain per row over a strip's columns instead of a sum of four), Newton's third law; every loop that serves strips --
    equal masses (S8), arbitrary masses (S9), per-particle softening with either (S12, S10), eps = 0 and the four-row /
    compiler-scheduled kernels (each tile's row sums added to the strip's in memory); 1, 2, 4 and 8 summation parts the same
    bits; fewer bytes of partial sums held; a column range that cuts a strip refused."""
    pos, vel = nb.plummer(n, seed=n + 7)
    rng = np.random.default_rng(n)
    eps_pp = rng.uniform(0.0, 0.02, n).astype(np.float32)
    rows = [(0, 128), (n // 2 - 64, n // 2 + 64), (n - 128, n)]

    def accel(strip, masses=None, eps=1e-3, pps=None, rpl=0, parts=0):
        p = pos.copy()
        if masses is not None:
            p[:, 3] = masses
        with nb.NBodySystem(n, split_len=2048) as s:
            s.set_force_mode("pair_once")
            s.set_strip_len(strip)
            s.set_rows_per_lane(rpl)
            s.set_summation_parts(parts)
            if pps is not None:
                s.set_particle_softening(pps)
            s.setParticlesPosition(p)
            s.setParticlesVelocity(np.zeros_like(p))
            s.step(1.0, eps)
            return s.download()[1][:, :3].astype(np.float64), s.partial_sum_bytes()

    random_masses = (pos[:, 3] * rng.uniform(0.5, 2.0, n)).astype(np.float32)
    cases = {"equal masses": {}, "arbitrary masses": {"masses": random_masses}, "pps, equal masses": {"pps": eps_pp},
             "pps, arbitrary masses": {"pps": eps_pp, "masses": random_masses}, "eps = 0": {"eps": 0.0},
             "four-row loops": {"rpl": 4}, "pps, eps = 0": {"pps": eps_pp, "eps": 0.0}}
    for name, kw in cases.items():
        a4, held4 = accel(4, **kw)                                # strips of four (automatic from 1024 splits on)
        a1, held1 = accel(1, **kw)
        p = pos.copy()
        if "masses" in kw:
            p[:, 3] = kw["masses"]
        for lo, hi in rows:
            want = (oracle_mod.accel_f64_pps(p, kw["pps"], kw.get("eps", 1e-3), i0=lo, i1=hi) if "pps" in kw
                    else oracle_mod.accel_f64(p, i0=lo, i1=hi, eps=kw.get("eps", 1e-3)))
            assert np.linalg.norm(a4[lo:hi] - want) / np.linalg.norm(want) < TOL, name
        assert np.linalg.norm(a4 - a1) / np.linalg.norm(a1) < 1e-6 and not np.array_equal(a4, a1), name
        m = p[:, 3].astype(np.float64)
        net = (m[:, None] * a4).sum(0)
        assert np.all(np.abs(net) < 1e-5 * (m[:, None] * np.abs(a4)).sum(0)), name
        assert held4 < 0.7 * held1, (name, held4, held1)          # rows: a quarter; columns: what they were
    base, _ = accel(4)
    for parts in (1, 2, 4, 8):
        assert np.array_equal(accel(4, parts=parts)[0], base), parts
    # the automatic strip length below 1024 splits is two (a rank of eight's share of the pass keeps short workgroups)
    a2, held2 = accel(2)
    auto, held_auto = accel(0)
    assert np.array_equal(auto, a2) and held_auto == held2 and held2 < 0.85 * accel(1)[1]   # 32 splits: 16 + 11 slots against 16 + 17
    assert np.linalg.norm(a2 - base) / np.linalg.norm(base) < 1e-6
    with nb.NBodySystem(n, split_len=2048) as s:
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        with pytest.raises(nb.NBodyError):
            s.forces(0, 3 * 2048, 1e-3)                              # one and a half strips of two column splits


def test_summation_parts_api(nb):
    """Changing the number of parts between steps is allowed (the plans are rebuilt); a column-range call after an all-columns
    call in several parts is refused (the earlier parts are already summed); bad values are refused."""
    n = 1 << 18
    pos, vel = nb.plummer(n, seed=23)
    with nb.NBodySystem(n, split_len=1024) as s:
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        with pytest.raises(nb.NBodyError):
            s.set_summation_parts(3)
        s.set_summation_parts(8)
        s.step(1e-3, 1e-3)
        s.set_summation_parts(2)
        s.step(1e-3, 1e-3)
        s.set_early_summation(False)
        s.step(1e-3, 1e-3)
        got = s.download()
        s.set_summation_parts(4)
        s.forces(0, n, 1e-3)
        with pytest.raises(nb.NBodyError):
            s.forces(0, 1024, 1e-3)
    with nb.NBodySystem(n, split_len=1024) as s:
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(3, 1e-3, 1e-3)
        want = s.download()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


# ---- diagnostics -----------------------------------------------------------------------------------

def test_energy_and_momentum_match_oracle(nb, oracle_mod):
    pos, vel = nb.plummer(6000, seed=15)
    with nb.NBodySystem(6000) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        e = s.energy(1e-3)
        m = s.momentum()
    assert np.allclose(e, oracle_mod.energy(pos, vel, 1e-3), rtol=1e-6)
    mo = oracle_mod.momentum(pos, vel)
    assert np.allclose(m, mo, rtol=1e-9, atol=1e-12)


def test_energy_drift_over_200_steps(nb):
    pos, vel = nb.plummer(16384, seed=16)
    with nb.NBodySystem(16384) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        e0 = s.energy(1e-2)
        s.step_n(200, 1e-3, 1e-2)
        e1 = s.energy(1e-2)
        m1 = s.momentum()
    assert abs(e1[2] - e0[2]) / abs(e0[2]) < 1e-4
    assert np.abs(m1[:3]).max() < 1e-5


# ---- BASELINE.json's full size, through size-independent properties ----------------------------------

def test_full_size_n1048576_properties(nb, oracle_mod):
    """N = 2^20 (configs[2]): one step; (1) a sample of rows vs the fp64 oracle, (2) sum_i m_i a_i = 0
    (Newton's third law over all 1.1e12 ordered pairs), (3) the state update is the oracle's kick-drift."""
    n = 1 << 20
    pos, vel = nb.plummer(n, seed=nb.CONFIG_SEED[3])
    with nb.NBodySystem(n) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(vel))
        s.step(1.0, 1e-3)                     # v_new = a
        acc = s.download()[1][:, :3].astype(np.float64)
    rows = np.concatenate([np.arange(0, 256), np.arange(n // 2 - 128, n // 2 + 128), np.arange(n - 256, n)])
    for lo, hi in ((0, 256), (n // 2 - 128, n // 2 + 128), (n - 256, n)):
        a64 = oracle_mod.accel_f64(pos, i0=lo, i1=hi, eps=1e-3)
        assert np.linalg.norm(acc[lo:hi] - a64) / np.linalg.norm(a64) < TOL
    assert rows.size == 768
    m = pos[:, 3].astype(np.float64)
    net = (m[:, None] * acc).sum(0)
    scale = (m[:, None] * np.abs(acc)).sum(0)
    assert np.all(np.abs(net) < 1e-5 * scale)


@pytest.fixture(scope="module")
def plummer_4m(nb):
    return nb.plummer(1 << 22, seed=nb.CONFIG_SEED[5])


def test_config5_size_n4194304_pair_once_whole_step(nb, oracle_mod, plummer_4m):
    """N = 2^22 in the pair-once mode on ONE GPU: 8.8e12 pair evaluations, 137 GB of partial sums whose offsets pass
    2^32 entries.  Sampled rows against the fp64 oracle and Newton's third law over all bodies."""
    n = 1 << 22
    pos, _ = plummer_4m
    with nb.NBodySystem(n, split_len=nb.pair_once_split_len(n)) as s:
        assert s.split_len == 2048
        s.set_force_mode("pair_once")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(pos))
        s.step(1.0, 1e-3)                     # v_new = a
        acc = s.download()[1][:, :3].astype(np.float64)
    for lo in (0, n // 2 - 32, 3 * (n // 4) + 1000, n - 64):
        a64 = oracle_mod.accel_f64(pos, i0=lo, i1=lo + 64, eps=1e-3)
        assert np.linalg.norm(acc[lo:lo + 64] - a64) / np.linalg.norm(a64) < TOL
    m = pos[:, 3].astype(np.float64)
    net = (m[:, None] * acc).sum(0)
    assert np.all(np.abs(net) < 1e-6 * (m[:, None] * np.abs(acc)).sum(0))


def test_config5_size_n4194304_one_rank_of_eight(nb, oracle_mod, plummer_4m):
    """N = 2^22 (configs[4]) as rank 3 of 8 sees it: its 524288 rows against all 4.2e6 columns, one step.  A sample of
    rows against the fp64 oracle, and the kick-drift of those rows."""
    from sharded_harness import shard_geometry
    n, world, rank = 1 << 22, 8, 3
    split_len = nb.default_split_len(n)
    n_padded, chunk = shard_geometry(n, world, split_len)
    assert n_padded == n and chunk == n // world
    pos, vel = plummer_4m
    lo = rank * chunk
    with nb.NBodySystem(n, row_lo=lo, row_count=chunk) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step(1e-3, 1e-3)
        new_pos, new_vel = s.download()
    new_pos = new_pos[lo:lo + chunk]
    assert new_vel.shape[0] == chunk
    for off in (0, chunk // 2, chunk - 64):
        a64 = oracle_mod.accel_f64(pos, i0=lo + off, i1=lo + off + 64, eps=1e-3)
        v_want = vel[lo + off:lo + off + 64, :3].astype(np.float64) + 1e-3 * a64
        x_want = pos[lo + off:lo + off + 64, :3].astype(np.float64) + 1e-3 * v_want
        got_a = (new_vel[off:off + 64, :3].astype(np.float64) - vel[lo + off:lo + off + 64, :3]) / 1e-3
        assert np.linalg.norm(got_a - a64) / np.linalg.norm(a64) < 1e-3   # a recovered through an fp32 velocity
        assert np.abs(new_vel[off:off + 64, :3] - v_want).max() / np.abs(v_want).max() < TOL
        assert np.abs(new_pos[off:off + 64, :3] - x_want).max() / np.abs(x_want).max() < TOL
    assert np.array_equal(new_pos[:, 3], pos[lo:lo + chunk, 3])


# ---- N1: the pair-once (symmetric) force kernel, the headline force mode ----------------------------------

def sym_run(nb, pos, vel, dt, eps, steps, mode, split_len=0):
    with nb.NBodySystem(pos.shape[0], split_len=split_len) as s:
        s.set_force_mode(mode)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, dt, eps)
        return s.download()


@pytest.mark.parametrize("n,split_len", [(16384, 1024), (20000, 1280), (5000, 1024), (3000, 256), (1500, 2048),
                                         (65536, 4096), (262144, 2048)])
def test_symmetric_mode_matches_oracle_and_one_sided(nb, oracle_mod, n, split_len):
    pos, vel = nb.uniform_cube(n, seed=200 + n, random_masses=True, speed=0.2) if n < 30000 else nb.plummer(n, seed=n)
    zero = np.zeros_like(vel)
    a_sym = sym_run(nb, pos, zero, 1.0, 1e-3, 1, "symmetric", split_len)[1][:, :3]     # dt = 1, v0 = 0: v = a
    a_one = sym_run(nb, pos, zero, 1.0, 1e-3, 1, "one_sided", split_len)[1][:, :3]
    rows = slice(0, n) if n <= 20000 else slice(n // 2 - 1024, n // 2 + 1024)
    a64 = oracle_mod.accel_f64(pos, i0=rows.start, i1=rows.stop, eps=1e-3)
    assert np.linalg.norm(a_sym[rows] - a64) / np.linalg.norm(a64) < TOL
    assert np.linalg.norm(a_sym - a_one) / np.linalg.norm(a_one) < 1e-6
    # Newton's third law holds to rounding for every pair, so the net force is tiny
    m = pos[:, 3:4].astype(np.float64)
    assert np.all(np.abs((m * a_sym).sum(0)) < 1e-6 * (m * np.abs(a_sym)).sum(0))
    p1, v1 = sym_run(nb, pos, vel, 1e-3, 1e-3, 3, "symmetric", split_len)
    p2, v2 = sym_run(nb, pos, vel, 1e-3, 1e-3, 3, "symmetric", split_len)
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2)                           # bit-reproducible
    pr, vr = sym_run(nb, pos, vel, 1e-3, 1e-3, 3, "one_sided", split_len)
    assert rel_state_error(p1, pr) < 1e-6 and rel_state_error(v1, vr) < 1e-6


@pytest.mark.parametrize("n", [200, 256, 1000, 5000, 20225, 66000])
def test_small_systems_tile_by_four_waves(nb, oracle_mod, n):
    """Round 4, 256-body splits (every system below 65 536 bodies): a tile is served by four waves, one 64-column group each, the
    diagonal tiles by the same kernel in the same launch (force_sym_quarter_kernel); 512-body splits (below 131 072 bodies; n =
    66 000 here): eight waves, two groups each.  Each WAVE decides whether its rows and its columns carry one mass: body sets whose species change inside a split, inside a 64-column group and on their boundaries;
    massless bodies; coincident bodies; eps = 0 (the guarded loop) and per-particle softening (their own loops); all against the
    fp64 truth and the one-sided kernels, the equal-mass path on and off, and two shards = one context, bit for bit."""
    rng = np.random.default_rng(n)
    L = nb.pair_once_split_len(n)                                                    # 256, or 512 from 65 536 bodies
    pos, vel = nb.plummer(n, seed=500 + n)
    cases = {"equal": pos.copy()}
    sp = pos.copy()
    for cut, m in zip(sorted(rng.integers(0, n + 1, 4)), (3.0, 0.25, 7.0, 0.5)):      # species in index order, boundaries anywhere
        sp[cut:, 3] = pos[cut:, 3] * m
    cases["species"] = sp
    rnd = pos.copy()
    rnd[:, 3] = rng.uniform(0.0, 2.0, n).astype(np.float32) / n
    rnd[rng.random(n) < 0.05, 3] = 0.0                                               # massless bodies
    k = rng.integers(0, n, size=max(1, n // 40))
    rnd[k, :3] = rnd[(k + 1) % n, :3]                                                # coincident bodies
    cases["random"] = rnd
    eps_pp = rng.uniform(0.0, 0.05, n).astype(np.float32)
    for name, state in cases.items():
        for eps, pps in ((1e-2, False), (0.0, False), (1e-2, True)):
            if eps == 0.0 and name != "random":
                continue
            acc = {}
            for mode, on in (("pair_once", True), ("pair_once", False), ("one_sided", True)):
                with nb.NBodySystem(n, split_len=L) as s:
                    s.set_force_mode(mode)
                    s.set_equal_mass_path(on)
                    if pps:
                        s.set_particle_softening(eps_pp)
                    s.setParticlesPosition(state)
                    s.setParticlesVelocity(np.zeros_like(vel))
                    s.step(1.0, eps)
                    acc[(mode, on)] = s.download()[1][:, :3].astype(np.float64)
            a64 = oracle_mod.accel_f64_pps(state, eps_pp, eps) if pps else oracle_mod.accel_f64(state, eps=eps)
            scale = np.linalg.norm(a64)
            err = {key: np.linalg.norm(a - a64) / scale for key, a in acc.items()}
            assert np.isfinite(acc[("pair_once", True)]).all(), (name, eps, pps)
            if eps > 0:
                assert max(err.values()) < TOL, (name, eps, pps, err)
            assert err[("pair_once", True)] <= max(1e-6, 3 * err[("one_sided", True)]), (name, eps, pps, err)
            assert np.linalg.norm(acc[("pair_once", True)] - acc[("pair_once", False)]) / scale < 1e-6
            m = state[:, 3:4].astype(np.float64)                                     # Newton's third law, pair by pair
            a = acc[("pair_once", True)]
            assert np.all(np.abs((m * a).sum(0)) <= 1e-6 * (m * np.abs(a)).sum(0) + 1e-30), (name, eps, pps)
    # steps are bit-reproducible, the graph replay included, and two row shards end with one context's bits
    state = cases["species"]
    runs = []
    for graph in (0, 1):
        with nb.NBodySystem(n, split_len=L) as s:
            s.set_force_mode("pair_once")
            s.set_graph_replay(graph)
            s.setParticlesPosition(state)
            s.setParticlesVelocity(vel)
            s.step_n(5, 1e-3, 1e-2)
            runs.append(s.download())
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    if n >= 4096:
        from n_body_problem_amd.multi import MultiGpuSystem
        with MultiGpuSystem(n, devices=[0, 0], force_mode="pair_once", transport="peer_copy", split_len=L) as m:
            m.set_state(state, vel)
            m.step_n(5, 1e-3, 1e-2)
            got = m.download()
            n_padded = m.n_padded
        pp, vv = np.zeros((n_padded, 4), np.float32), np.zeros((n_padded, 4), np.float32)
        pp[:n], vv[:n] = state, vel
        with nb.NBodySystem(n_padded, split_len=L) as s:
            s.set_force_mode("pair_once")
            s.setParticlesPosition(pp)
            s.setParticlesVelocity(vv)
            s.step_n(5, 1e-3, 1e-2)
            want = s.download()
        assert np.array_equal(got[0], want[0][:n]) and np.array_equal(got[1], want[1][:n])


def test_symmetric_mode_row_shards_with_a_manual_exchange(nb):
    """Two contexts in one process share the rows (4 + 4 groups); the column sums are exchanged by copying the slices
    each context wrote: bit-identical to the single context, for the pair-once summation order is fixed."""
    n, L = 16384, 512                                 # 32 splits, 8 groups of 4
    pos, vel = nb.plummer(n, seed=77)
    want = sym_run(nb, pos, vel, 1e-3, 1e-3, 2, "symmetric", L)
    a = nb.NBodySystem(n, row_lo=0, row_count=n // 2, split_len=L)
    b = nb.NBodySystem(n, row_lo=n // 2, row_count=n // 2, split_len=L)
    for s in (a, b):
        s.set_force_mode("symmetric")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
    assert a.sym_groups() == (0, 4, 4) and b.sym_groups() == (4, 4, 4)
    for _ in range(2):
        for s in (a, b):
            s.forces(s.row_lo, s.row_count, 1e-3)     # own columns first, then the rest, as the sharded host does
            s.forces_complement(s.row_lo, s.row_count, 1e-3)
            s.sym_reduce()
        a.colparts[4:8].copy_(b.colparts[4:8])
        b.colparts[0:4].copy_(a.colparts[0:4])
        for s in (a, b):
            s.update(1e-3)
        a.positions[n // 2:].copy_(b.positions[n // 2:])
        b.positions[:n // 2].copy_(a.positions[:n // 2])
    pa, va = a.download()
    pb, vb = b.download()
    assert np.array_equal(pa, want[0]) and np.array_equal(pb, want[0])
    assert np.array_equal(np.concatenate([va, vb]), want[1])
    with pytest.raises(nb.NBodyError):                # a shard cannot update before its sym_reduce
        a.forces(0, n, 1e-3)
        a.update(1e-3)
    a.close()
    b.close()
    with nb.NBodySystem(n, row_lo=L, row_count=4 * L, split_len=L) as s:   # rows must be whole groups
        with pytest.raises(nb.NBodyError):
            s.set_force_mode("symmetric")


def test_force_mode_can_be_switched_on_a_live_context(nb):
    n, L = 8192, 1024
    pos, vel = nb.plummer(n, seed=12)
    want = {m: sym_run(nb, pos, vel, 1e-3, 1e-3, 1, m, L) for m in ("one_sided", "symmetric")}
    with nb.NBodySystem(n, split_len=L) as s:
        for m in ("symmetric", "one_sided", "symmetric", "one_sided"):   # the partial-sum array grows on the way back
            s.set_force_mode(m)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step(1e-3, 1e-3)
            p, v = s.download()
            assert np.array_equal(p, want[m][0]) and np.array_equal(v, want[m][1]), m


def test_symmetric_mode_zero_softening_and_limits(nb):
    pos, vel = nb.uniform_cube(16384, seed=9, random_masses=True)
    pos[5] = pos[6]                                   # two coincident bodies
    a = sym_run(nb, pos, np.zeros_like(vel), 1.0, 0.0, 1, "symmetric", 1024)[1][:, :3]
    b = sym_run(nb, pos, np.zeros_like(vel), 1.0, 0.0, 1, "one_sided", 1024)[1][:, :3]
    assert np.all(np.isfinite(a)) and np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-6
    with nb.NBodySystem(1 << 20) as s:                # default split_len = 8192 > 4096
        with pytest.raises(nb.NBodyError):
            s.set_force_mode("symmetric")


# ---- N4: kick-drift-kick integrator ----------------------------------------------------------------------------

def test_kdk_matches_oracle_and_conserves_energy_better(nb, oracle_mod):
    pos, vel = nb.plummer(4096, seed=44)
    with nb.NBodySystem(4096) as s:
        s.set_integrator("kdk")
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        e0 = s.energy(2e-2)
        s.step_n(20, 1e-2, 2e-2)
        p, v = s.download()
        e_kdk = s.energy(2e-2)
    pr, vr = oracle_mod.step_kdk_f32(pos, vel, 1e-2, 2e-2, nsteps=20)
    assert rel_state_error(p, pr) < TOL and rel_state_error(v, vr) < TOL
    with nb.NBodySystem(4096) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(20, 1e-2, 2e-2)
        e_kd = s.energy(2e-2)
    assert abs(e_kdk[2] - e0[2]) < abs(e_kd[2] - e0[2])
    with nb.NBodySystem(4096) as s:                       # the pieces need their prerequisites
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        with pytest.raises(nb.NBodyError):
            s.kdk_kick_drift(1e-2)
        with pytest.raises(nb.NBodyError):
            s.kdk_kick(1e-2)


# ---- per-particle softening (the vel.w the reference loads and never reads) --------------

def pps_accel(nb, pos, eps_pp, eps, **kw):
    n = pos.shape[0]
    with nb.NBodySystem(n, **kw) as s:
        s.set_particle_softening(eps_pp)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(pos))
        s.step(1.0, eps)
        return s.download()[1][:, :3]


@pytest.mark.parametrize("n,eps", [(3000, 1e-3), (4097, 0.0)])
def test_per_particle_softening_matches_oracle(nb, oracle_mod, n, eps):
    pos, _ = nb.plummer(n, seed=41)
    rng = np.random.default_rng(41)
    eps_pp = rng.uniform(0.0, 0.05, n).astype(np.float32)
    eps_pp[::7] = 0.0   # unsoftened bodies among softened ones
    pos[11] = pos[10]   # a coincident pair: softened unless both lengths and eps are 0
    got = pps_accel(nb, pos, eps_pp, eps)
    want = oracle_mod.accel_f64_pps(pos, eps_pp, eps)
    assert np.isfinite(got).all()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() / scale <= TOL
    # equal lengths e are the global softening sqrt(eps^2 + 2 e^2), and 0 lengths are the plain kernel bit for bit
    assert np.array_equal(pps_accel(nb, pos, np.zeros(n, np.float32), 1e-3), gpu_accel(nb, pos, 1e-3, rpl=-4))
    e = np.float32(0.01)
    uni = pps_accel(nb, pos, np.full(n, e, np.float32), 0.0)
    glob = gpu_accel(nb, pos, float(np.sqrt(2.0) * e))
    assert np.abs(uni - glob).max() / np.abs(glob).max() <= TOL


@pytest.mark.parametrize("n,split_len,eps,masses", [(5000, 256, 1e-3, "equal"), (5000, 256, 0.0, "random"), (7000, 1024, 1e-3, "random"),
                                                    (3001, 512, 0.0, "equal"), (20225, 0, 1e-2, "species")])
def test_per_particle_softening_hand_scheduled_one_sided_loops(nb, oracle_mod, n, split_len, eps, masses):
    """Round 3: the packed one-sided loops (1024-row workgroups, rows_per_lane 4; one wave per workgroup, 41 -- what small
    systems take by default) carry the softening term themselves.  Same FMA chain per row as the compiler-allocated kernel
    (rows_per_lane -4): equal bits, for one-tile and longer splits, with and without the zero-distance guard, equal and
    arbitrary masses."""
    pos, _ = nb.plummer(n, seed=n)
    rng = np.random.default_rng(n)
    if masses == "random":
        pos[:, 3] = rng.uniform(0.0, 2.0 / n, n).astype(np.float32)
    elif masses == "species":
        pos[n // 3:, 3] *= np.float32(3.0)
    eps_pp = rng.uniform(0.0, 0.05, n).astype(np.float32)
    eps_pp[::5] = 0.0
    pos[21] = pos[20]
    got = {}
    for rpl in (-4, 4, 41, 0):
        with nb.NBodySystem(n, split_len=split_len) as s:
            s.set_rows_per_lane(rpl)
            s.set_particle_softening(eps_pp)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(pos))
            s.step(1.0, eps)
            got[rpl] = s.download()[1][:, :3]
    assert np.isfinite(got[-4]).all()
    for rpl in (4, 41, 0):
        assert np.array_equal(got[rpl], got[-4]), rpl
    want = oracle_mod.accel_f64_pps(pos, eps_pp, eps)
    assert np.abs(got[0] - want).max() / np.abs(want).max() <= TOL


def test_per_particle_softening_shards_and_energy(nb, oracle_mod):
    n, split_len = 6000, 512
    pos, vel = nb.plummer(n, seed=42)
    eps_pp = np.random.default_rng(42).uniform(0.0, 0.03, n).astype(np.float32)
    whole = pps_accel(nb, pos, eps_pp, 1e-3, split_len=split_len)
    import torch
    eps_dev = torch.from_numpy(eps_pp).cuda()
    parts = []
    for lo, cnt in [(0, 2048), (2048, 2048), (4096, n - 4096)]:  # rows sharded; every shard sees every body's length
        with nb.NBodySystem(n, row_lo=lo, row_count=cnt, split_len=split_len) as s:
            s.set_particle_softening(eps_dev)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros((cnt, 4), np.float32))
            s.step(1.0, 1e-3)
            parts.append(s.download()[1][:, :3])
    assert np.array_equal(np.concatenate(parts), whole)
    with nb.NBodySystem(n) as s:
        s.set_particle_softening(eps_pp)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        e = s.energy(1e-3)
        s.set_particle_softening(None)
        e_plain = s.energy(1e-3)
    assert np.isclose(e[1], oracle_mod.potential_pps(pos, eps_pp, 1e-3), rtol=1e-6)
    assert np.allclose(e_plain, oracle_mod.energy(pos, vel, 1e-3), rtol=1e-6)


@pytest.mark.parametrize("eps", [1e-3, 0.0])
def test_per_particle_softening_in_the_pair_once_mode(nb, oracle_mod, eps):
    """eps_ij^2 is symmetric in the pair, so the pair-once kernels take it too: against the fp64 oracle, against the
    one-sided kernel, and bit-identical when the rows are shared by two contexts."""
    n, L = 12000, 1024                                # 12 splits (the last one ragged), 8 groups of 2
    pos, _ = nb.plummer(n, seed=43)
    eps_pp = np.random.default_rng(43).uniform(0.0, 0.05, n).astype(np.float32)
    eps_pp[::5] = 0.0
    pos[21] = pos[20]                                 # a coincident pair, inside one tile
    pos[5000] = pos[100]                              # and one across tiles
    sym = pps_accel_mode(nb, pos, eps_pp, eps, "symmetric", L)
    one = pps_accel_mode(nb, pos, eps_pp, eps, "one_sided", L)
    want = oracle_mod.accel_f64_pps(pos, eps_pp, eps)
    assert np.isfinite(sym).all()
    assert np.abs(sym - want).max() / np.abs(want).max() <= TOL
    assert np.linalg.norm(sym - one) / np.linalg.norm(one) < 1e-6
    zero = pps_accel_mode(nb, pos, np.zeros(n, np.float32), 1e-3, "symmetric", L)
    plain = sym_run(nb, pos, np.zeros_like(pos), 1.0, 1e-3, 1, "symmetric", L)[1][:, :3]
    assert np.abs(zero - plain).max() / np.abs(plain).max() < 3e-6    # another kernel (and m x sum instead of sum of m x term): rounding
    # two contexts sharing the rows (groups 0-3 and 4-7 = splits 0-7 and 8-11), column sums copied by hand
    import torch
    half = 8 * L
    a = nb.NBodySystem(n, row_lo=0, row_count=half, split_len=L)
    b = nb.NBodySystem(n, row_lo=half, row_count=n - half, split_len=L)
    for s in (a, b):
        s.set_force_mode("symmetric")
        s.set_particle_softening(eps_pp)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros((s.row_count, 4), np.float32))
        s.forces(0, n, eps)
        s.sym_reduce()
    a.colparts[4:8].copy_(b.colparts[4:8])
    b.colparts[0:4].copy_(a.colparts[0:4])
    for s in (a, b):
        s.update(1.0)
    got = np.concatenate([a.download()[1][:, :3], b.download()[1][:, :3]])
    a.close()
    b.close()
    assert np.array_equal(got, sym)


@pytest.mark.parametrize("n,L", [(20000, 2048), (9000, 1024), (4096, 1024), (9000, 512), (1536, 512), (9000, 256), (5000, 768),
                                 (700, 256), (6000, 1280)])
def test_per_particle_softening_eight_row_loop(nb, oracle_mod, n, L):
    """Round 3: with eps > 0 the pair-once tiles take hand-scheduled loops that carry the softening term -- splits of whole 512
    bodies the eight-row loop (S10), the others the four-row loop (S11) -- instead of the compiler-scheduled kernel
    (rows_per_lane 4 keeps it).  Splits of whole 512 bodies: the eight-row loop with eps_j^2 staged beside
    the columns (S10_GROUP_LOOP): against the fp64 oracle, the one-sided kernel and the compiler-scheduled pair-once kernel
    (rows_per_lane 4 keeps it)."""
    rng = np.random.default_rng(n + L)
    pos, _ = nb.plummer(n, seed=47)
    pos[:, 3] = rng.uniform(0.0, 2.0 / n, n).astype(np.float32)      # arbitrary masses, some of them tiny
    eps_pp = rng.uniform(0.0, 0.05, n).astype(np.float32)
    eps_pp[::7] = 0.0
    pos[33] = pos[32]
    pos[n - 1, :3] = pos[10, :3]
    sym = pps_accel_mode(nb, pos, eps_pp, 1e-3, "symmetric", L)
    one = pps_accel_mode(nb, pos, eps_pp, 1e-3, "one_sided", L)
    want = oracle_mod.accel_f64_pps(pos, eps_pp, 1e-3)
    assert np.isfinite(sym).all()
    assert np.abs(sym - want).max() / np.abs(want).max() <= TOL
    assert np.linalg.norm(sym - want) / np.linalg.norm(want) < 2e-6
    assert np.linalg.norm(sym - one) / np.linalg.norm(one) < 1e-6
    plain = pps_accel_mode(nb, pos, eps_pp, 1e-3, "symmetric", L, rpl=4)
    assert np.linalg.norm(sym - plain) / np.linalg.norm(plain) < 1e-6
    assert not np.array_equal(sym, plain)                                # two kernels: m x sum against sum of m x term
    # summation parts and the step graph leave the bits alone
    with nb.NBodySystem(n, split_len=L) as s:
        s.set_force_mode("symmetric")
        s.set_summation_parts(4)
        s.set_particle_softening(eps_pp)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(pos))
        s.step(1.0, 1e-3)
        assert np.array_equal(s.download()[1][:, :3], sym)


@pytest.mark.parametrize("n,L,species", [(20000, 2048, 1), (9000, 1024, 1), (9000, 512, 1), (24000, 1024, 3)])
def test_per_particle_softening_equal_mass_tiles_take_their_own_loop(nb, oracle_mod, n, L, species):
    """Round 4: under per-particle softening a tile whose two splits each carry ONE mass (every tile of an equal-mass body
    set -- the benchmark's sphere with vel.w in use -- and most tiles of a few-species set) runs S12_GROUP_LOOP: the eight-row
    loop with the softening term and WITHOUT the masses (7.5 packed instructions + 1 transcendental per pair instead of
    S10's 8.5 + 1).  Against the fp64 oracle, the one-sided kernel, and the same library with the equal-mass path off (S10 on
    every tile): equal to rounding, not bit for bit (m x sum against sum of m x term)."""
    rng = np.random.default_rng(n + L + species)
    pos, _ = nb.plummer(n, seed=48)
    if species > 1:                                   # a few-species set stored species by species: some tiles mixed
        masses = np.float32(1.0 / n) * np.array([1.0, 4.0, 0.25], dtype=np.float32)
        pos[:, 3] = masses[np.minimum(np.arange(n) * species // n, species - 1)]
    eps_pp = rng.uniform(0.0, 0.05, n).astype(np.float32)
    eps_pp[::5] = 0.0
    pos[33] = pos[32]

    def accel(equal_mass_path):
        with nb.NBodySystem(n, split_len=L) as s:
            s.set_force_mode("pair_once")
            s.set_equal_mass_path(equal_mass_path)
            s.set_particle_softening(eps_pp)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(pos))
            s.step(1.0, 1e-3)
            return s.download()[1][:, :3]
    fast, general = accel(True), accel(False)
    want = oracle_mod.accel_f64_pps(pos, eps_pp, 1e-3)
    one = pps_accel_mode(nb, pos, eps_pp, 1e-3, "one_sided", L)
    assert np.isfinite(fast).all()
    assert np.abs(fast - want).max() / np.abs(want).max() <= TOL
    assert np.linalg.norm(fast - want) / np.linalg.norm(want) < 2e-6
    assert np.linalg.norm(fast - one) / np.linalg.norm(one) < 1e-6
    assert np.linalg.norm(fast - general) / np.linalg.norm(general) < 1e-6
    assert not np.array_equal(fast, general)          # two loops


def pps_accel_mode(nb, pos, eps_pp, eps, mode, split_len, rpl=0):
    n = pos.shape[0]
    with nb.NBodySystem(n, split_len=split_len) as s:
        s.set_force_mode(mode)
        if rpl:
            s.set_rows_per_lane(rpl)
        s.set_particle_softening(eps_pp)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(np.zeros_like(pos))
        s.step(1.0, eps)
        return s.download()[1][:, :3]


def test_a_cached_step_graph_never_outlives_the_buffers_it_launches_on(nb):
    """the captured step bakes in the partial-sum arrays; a mode switch that reallocates them (the one-sided
    array is larger than the pair-once ones at this size) used to leave a graph that replayed on freed memory.  Pair-once
    -> one-sided -> pair-once with the same dt and softening must equal the eagerly launched loop bit for bit."""
    n = 4096
    pos, vel = nb.plummer(n, seed=17)
    out = {}
    for replay in (1, 0):
        with nb.NBodySystem(n, split_len=256) as s:
            s.set_graph_replay(replay)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            for mode in ("pair_once", "one_sided", "pair_once", "one_sided"):
                s.set_force_mode(mode)
                s.step_n(5, 1e-3, 1e-2)
            s.set_summation_parts(1)
            s.set_force_mode("pair_once")
            s.step_n(5, 1e-3, 1e-2)
            out[replay] = s.download()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])


def test_c_abi_auto_mode_lands_on_the_fast_kernels(nb, oracle_mod):
    """nbody_create_auto / nbody_set_force_mode(NBODY_FORCE_AUTO) from a C host's point of view:
    one call gives the force mode -- and the split length it needs -- that initialize(force_mode="auto") gives Python."""
    from n_body_problem_amd import _lib
    lib = _lib.load()
    for n, want_mode in ((1000, 1), (20225, 1), (32768, 1), (1 << 18, 1)):   # round 4: the pair-once kernels at every size
        pos, vel = nb.plummer(n, seed=18)
        want_len = nb.pair_once_split_len(n) if want_mode else nb.default_split_len(n)
        results = []
        for how in ("create_auto", "set_force_mode"):
            ctx = ctypes.c_void_p(None)
            if how == "create_auto":
                assert lib.nbody_create_auto(ctypes.byref(ctx), 0, n) == 0, lib.nbody_last_error(None)
            else:
                assert lib.nbody_create(ctypes.byref(ctx), 0, n) == 0
                assert lib.nbody_set_positions(ctx, pos.ctypes.data_as(ctypes.c_void_p)) == 0     # buffers survive the re-split
                assert lib.nbody_set_force_mode(ctx, 2) == 0, lib.nbody_last_error(ctx)
            assert lib.nbody_force_mode(ctx) == want_mode and lib.nbody_split_len(ctx) == want_len
            assert lib.nbody_set_positions(ctx, pos.ctypes.data_as(ctypes.c_void_p)) == 0
            assert lib.nbody_set_velocities(ctx, vel.ctypes.data_as(ctypes.c_void_p)) == 0
            assert lib.nbody_step_n(ctx, 2, 1e-3, 1e-3) == 0, lib.nbody_last_error(ctx)
            p, v = np.empty_like(pos), np.empty_like(vel)
            assert lib.nbody_download(ctx, p.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)) == 0
            assert lib.nbody_destroy(ctx) == 0
            results.append((p, v))
        assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
        p2, v2 = run_gpu(nb, pos, vel, 1e-3, 1e-3, 2, "pair_once" if want_mode else "one_sided")
        assert np.array_equal(results[0][0], p2) and np.array_equal(results[0][1], v2)
        if n <= 65536:
            pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-3, nsteps=2)
            assert rel_state_error(results[0][0], pr) < TOL and rel_state_error(results[0][1], vr) < TOL
    shard = ctypes.c_void_p(None)
    assert lib.nbody_create_shard(ctypes.byref(shard), 0, 8192, 0, 4096, 256) == 0
    assert lib.nbody_set_force_mode(shard, 2) == _lib.NBODY_ERR_INVALID            # a shard's split length is part of the sharding
    assert lib.nbody_destroy(shard) == 0


@pytest.mark.parametrize("n,split_len", [(16384, 1024), (16384, 2048), (21000, 1024), (12288, 2048)])
def test_eight_row_loop_for_arbitrary_masses(nb, oracle_mod, n, split_len):
    """Round 3: tiles of splits of whole 1024 bodies run eight rows per lane whatever their masses -- the equal-mass loop or
    the loop for arbitrary masses (S9_GROUP_LOOP), both in one kernel allocated for three waves per SIMD (2048-body splits:
    four waves per workgroup, 1024: two).  Random masses, massless bodies and a ragged last split: against the fp64 truth;
    against the four-row loops (nbody_set_rows_per_lane(4)) to rounding; the equal-mass switch changes nothing when no tile is
    one-mass; two shards = one context bit for bit; and run to run."""
    from n_body_problem_amd.multi import MultiGpuSystem
    pos, vel = nb.plummer(n, seed=41)
    rng = np.random.default_rng(41)
    pos[:, 3] *= rng.uniform(0.2, 3.0, n).astype(np.float32)
    pos[rng.integers(0, n, 50), 3] = 0.0
    acc, state = {}, {}
    for how in ("eight", "eight, equal-mass path off", "four", "eight again"):
        with nb.NBodySystem(n, split_len=split_len) as s:
            s.set_force_mode("pair_once")
            s.set_equal_mass_path(how != "eight, equal-mass path off")
            s.set_rows_per_lane(4 if how == "four" else 0)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(np.zeros_like(vel))
            s.step(1.0, 1e-2)
            acc[how] = s.download()[1][:, :3].astype(np.float64)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(3, 1e-3, 1e-2)
            state[how] = s.download()
    for other in ("eight, equal-mass path off", "eight again"):
        assert np.array_equal(acc[other], acc["eight"]), other
        assert np.array_equal(state[other][0], state["eight"][0]) and np.array_equal(state[other][1], state["eight"][1]), other
    a64 = oracle_mod.accel_f64(pos, eps=1e-2)
    scale = np.linalg.norm(a64)
    assert np.linalg.norm(acc["eight"] - a64) / scale < TOL and np.linalg.norm(acc["four"] - a64) / scale < TOL
    assert 0 < np.linalg.norm(acc["eight"] - acc["four"]) / scale < 1e-6  # another association of the same sums
    mass = pos[:, 3].astype(np.float64)
    net = (mass[:, None] * acc["eight"]).sum(0)
    assert np.all(np.abs(net) < 1e-5 * (mass[:, None] * np.abs(acc["eight"])).sum(0))   # every pair once, to both bodies
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-2, nsteps=3)
    assert rel_state_error(state["eight"][0], pr) < TOL and rel_state_error(state["eight"][1], vr) < TOL
    with MultiGpuSystem(n, devices=[0, 0], force_mode="pair_once", transport="peer_copy", split_len=split_len) as m:
        m.set_state(pos, vel)
        m.step_n(3, 1e-3, 1e-2)
        p, v = m.download()
        n_padded = m.n_padded
    if n_padded == n:
        assert np.array_equal(p, state["eight"][0]) and np.array_equal(v, state["eight"][1])
    else:                                                        # padded to whole groups: other split counts, the same physics
        assert rel_state_error(p, state["eight"][0]) < 1e-6 and rel_state_error(v, state["eight"][1]) < 1e-6
