"""GPU: the C++ host (host/nbody_run) drives the same C ABI as the Python mirror -- same bits -- and its
snapshot dump/resume continues a run exactly."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_cli(*args):
    from n_body_problem_amd import build
    exe = build.build_host()
    res = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    return res.stdout


def test_cli_matches_python_mirror_and_resumes_exactly(tmp_path):
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    pos, vel = nb.plummer(5000, seed=91)
    start = str(tmp_path / "start.nbs")
    ds.save_snapshot(start, pos, vel, step=0, time=0.0)
    with nb.NBodySystem(5000) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(5, 1e-3, 1e-3)
        want_p, want_v = s.download()
    out = run_cli("--resume", start, "--steps", 5, "--dt", 1e-3, "--softening", 1e-3, "--energy-every", 5,
                  "--final", tmp_path / "five.nbs")
    assert "interactions/s" in out and "dE/E0" in out
    p, v, step, time = ds.load_snapshot(str(tmp_path / "five.nbs"))
    assert step == 5 and time == pytest.approx(5e-3)
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)
    # 3 steps, dump, resume for 2 more: identical to 5 in one go
    run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--dump-every", 3, "--dump-prefix",
            tmp_path / "run")
    run_cli("--resume", tmp_path / "run_000003.nbs", "--steps", 2, "--dt", 1e-3, "--softening", 1e-3, "--final",
            tmp_path / "resumed.nbs")
    p2, v2, step2, _ = ds.load_snapshot(str(tmp_path / "resumed.nbs"))
    assert step2 == 5 and np.array_equal(p2, want_p) and np.array_equal(v2, want_v)


def test_cli_reads_the_reference_formats(tmp_path):
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    pos, vel = nb.uniform_cube(300, seed=5, random_masses=True, speed=0.1)
    f = str(tmp_path / "g.bin")
    ds.write_tipsy(f, pos, vel, ndark=100)
    run_cli("--file", f, "--steps", 2, "--final", tmp_path / "o.nbs", "--pad-reference")   # reference dt / softening
    p, v, _, _ = ds.load_snapshot(str(tmp_path / "o.nbs"))
    assert p.shape[0] == nb.padded_count(300)
    with nb.NBodySystem(300) as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(2, nb.TIME_TICK, nb.SOFTENING_VERSION3)
        want_p, want_v = s.download()
    err = np.abs(p[:300, :3] - want_p[:, :3]).max() / np.abs(want_p[:, :3]).max()
    assert err < 1e-6 and np.array_equal(v[:300, 3], vel[:, 3])


def test_cli_particle_softening_from_the_velocity_records(tmp_path):
    """--particle-softening uses the eps column the reference's loaders fill (kernel.cu:223) as per-particle lengths."""
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    pos, vel = nb.plummer(3000, seed=92)
    vel[:, 3] = np.random.default_rng(92).uniform(0.0, 0.05, 3000).astype(np.float32)
    start = str(tmp_path / "start.nbs")
    ds.save_snapshot(start, pos, vel, step=0, time=0.0)
    run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--particle-softening", "--final",
            tmp_path / "out.nbs")
    p, v, _, _ = ds.load_snapshot(str(tmp_path / "out.nbs"))
    with nb.NBodySystem(3000) as s:
        s.set_particle_softening(vel[:, 3])
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(3, 1e-3, 1e-3)
        want_p, want_v = s.download()
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)
    with nb.NBodySystem(3000) as s:   # and it is not the plain run
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(3, 1e-3, 1e-3)
        assert not np.array_equal(s.download()[1], want_v)


def test_cli_morton_flag_matches_the_python_mirror_and_keeps_the_files_order(tmp_path):
    """--morton stores the bodies along the curve (nbody_morton_order in the C++ host) and writes snapshots in the file's
    order: bit-identical to NBodySystem(body_order="morton"), one device and two shards on it."""
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    pos, vel = nb.plummer(6000, seed=95)
    pos[::2, 3] *= 2.0
    vel[:, 3] = np.random.default_rng(95).uniform(0.0, 0.05, 6000).astype(np.float32)
    start = str(tmp_path / "start.nbs")
    ds.save_snapshot(start, pos, vel, step=0, time=0.0)
    with nb.NBodySystem(6000, body_order="morton") as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.set_particle_softening(vel[:, 3])
        s.step_n(3, 1e-3, 1e-3)
        want_p, want_v = s.download()
    run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--particle-softening", "--morton", "--final",
            tmp_path / "one.nbs")
    p, v, _, _ = ds.load_snapshot(str(tmp_path / "one.nbs"))
    assert np.array_equal(p, want_p) and np.array_equal(v, want_v)
    # the layout refreshed every 4 steps: the schedule of set_reorder_period / nbody_multi_set_reorder_period
    with nb.NBodySystem(6000, body_order="morton") as s:
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.set_particle_softening(vel[:, 3])
        s.set_reorder_period(4)
        s.step_n(10, 5e-3, 1e-3)
        want10_p, want10_v = s.download()
    run_cli("--resume", start, "--steps", 10, "--dt", 5e-3, "--softening", 1e-3, "--particle-softening", "--morton",
            "--reorder-every", 4, "--final", tmp_path / "ten.nbs")
    p10, v10, _, _ = ds.load_snapshot(str(tmp_path / "ten.nbs"))
    assert np.array_equal(p10, want10_p) and np.array_equal(v10, want10_v)
    run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--particle-softening", "--morton",
            "--devices", "0,0", "--peer-copy", "--final", tmp_path / "two.nbs")
    p2, v2, _, _ = ds.load_snapshot(str(tmp_path / "two.nbs"))
    assert np.array_equal(p2[:, 3], pos[:, 3]) and np.array_equal(v2[:, 3], vel[:, 3])
    assert np.abs(p2[:, :3] - want_p[:, :3]).max() <= 1e-6 * np.abs(want_p[:, :3]).max()   # other split boundaries: rounding


def test_cli_pair_once_and_kdk_flags_match_the_python_mirror(tmp_path):
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    pos, vel = nb.plummer(6000, seed=93)
    start = str(tmp_path / "start.nbs")
    ds.save_snapshot(start, pos, vel, step=0, time=0.0)
    for flags, mode, integ, split_len in ((["--pair-once"], "pair_once", "kick_drift", nb.pair_once_split_len(6000)),
                                          (["--kdk"], "one_sided", "kdk", 0),
                                          (["--pair-once", "--kdk"], "pair_once", "kdk", nb.pair_once_split_len(6000))):
        final = str(tmp_path / ("out_" + "_".join(f.strip("-") for f in flags) + ".nbs"))
        run_cli("--resume", start, "--steps", 4, "--dt", 1e-3, "--softening", 1e-3, "--final", final, *flags)
        p, v, _, _ = ds.load_snapshot(final)
        with nb.NBodySystem(6000, split_len=split_len) as s:
            s.set_force_mode(mode)
            s.set_integrator(integ)
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(4, 1e-3, 1e-3)
            want_p, want_v = s.download()
        assert np.array_equal(p, want_p) and np.array_equal(v, want_v), flags


def test_cli_devices_runs_the_library_owned_multi_gpu_step(tmp_path):
    """nbody_run --devices: the C++ host drives nbody_multi_* (both ranks on cuda:0, peer copies standing in for RCCL,
    which refuses duplicate devices) and ends with the bits of the single-context run on the same padded system."""
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    from n_body_problem_amd.multi import geometry
    n = 6000
    pos, vel = nb.plummer(n, seed=92)
    start = str(tmp_path / "start.nbs")
    ds.save_snapshot(start, pos, vel, step=0, time=0.0)
    for flags, mode in ((["--pair-once"], "pair_once"), ([], "one_sided"), (["--ring", "--kdk"], "one_sided")):
        out = run_cli("--resume", start, "--steps", 4, "--dt", 1e-3, "--softening", 1e-3, "--energy-every", 2, "--devices", "0,0",
                      "--peer-copy", *flags, "--final", tmp_path / "multi.nbs")
        assert "replicas identical: yes" in out and "ranks = 2" in out and "dE/E0" in out
        p, v, step, _ = ds.load_snapshot(str(tmp_path / "multi.nbs"))
        n_padded, _, split_len = geometry(n, 2, mode)
        pp = np.zeros((n_padded, 4), np.float32)
        vv = np.zeros((n_padded, 4), np.float32)
        pp[:n], vv[:n] = pos, vel
        with nb.NBodySystem(n_padded, split_len=split_len) as s:
            s.set_force_mode(mode)
            s.set_integrator("kdk" if "--kdk" in flags else "kick_drift")
            s.setParticlesPosition(pp)
            s.setParticlesVelocity(vv)
            s.step_n(4, 1e-3, 1e-3)
            want_p, want_v = s.download()
        assert step == 4 and np.array_equal(p, want_p[:n]) and np.array_equal(v, want_v[:n]), flags


def test_config5_dry_run_two_ranks_on_one_gpu_with_snapshots(tmp_path):
    """BASELINE configs[4] at reduced length: N = 4 194 304 in the pair-once mode, rows sharded over two ranks (both on
    cuda:0, peer copies in RCCL's place), 20 steps with the energy at both ends and a snapshot every 10 steps; the run
    resumed from the step-10 snapshot ends with the bits of the uninterrupted one.  On a node the 1000-step run is the
    same command with --devices 0,...,7 --steps 1000 --energy-every 100 (RCCL instead of --peer-copy)."""
    import re
    from n_body_problem_amd import datasets as ds
    common = ["--devices", "0,0", "--peer-copy", "--pair-once", "--dt", 1e-3, "--softening", 1e-2]
    out = run_cli("--plummer", 1 << 22, "--seed", 0x5EED0005, *common, "--steps", 20, "--energy-every", 20, "--dump-every", 10,
                  "--dump-prefix", tmp_path / "c5", "--final", tmp_path / "straight.nbs")
    assert "ranks = 2" in out and "rows per rank = 2097152" in out and "split = 2048" in out
    assert "replicas identical: yes" in out
    drift = [float(x) for x in re.findall(r"dE/E0 = ([-+0-9.e]+)", out)]
    assert len(drift) == 1 and abs(drift[0]) < 1e-5, out
    rate = float(re.search(r"([0-9.e+]+) interactions/s", out).group(1))
    assert rate > 2e12                                       # two ranks sharing one GPU still run at the one-GPU rate
    out2 = run_cli("--resume", tmp_path / "c5_000010.nbs", *common, "--steps", 10, "--final", tmp_path / "resumed.nbs")
    assert "replicas identical: yes" in out2
    p, v, step, time = ds.load_snapshot(str(tmp_path / "straight.nbs"))
    p2, v2, step2, time2 = ds.load_snapshot(str(tmp_path / "resumed.nbs"))
    assert step == step2 == 20 and time == pytest.approx(time2) and p.shape == (1 << 22, 4)
    assert np.array_equal(p, p2) and np.array_equal(v, v2)


def test_cli_auto_flag_is_the_librarys_choice_of_force_mode(tmp_path):
    """--auto = nbody_create_auto (one context) / NBODY_FORCE_AUTO in nbody_multi_config (--devices): the pair-once
    kernels at every size since round 4 (NBODY_PAIR_ONCE_MIN_BODIES = 0) -- the same bits as initialize(force_mode="auto")."""
    import n_body_problem_amd as nb
    from n_body_problem_amd import datasets as ds
    from n_body_problem_amd.multi import MultiGpuSystem
    for n, want in ((6000, "pair-once"), (32768, "pair-once")):
        pos, vel = nb.plummer(n, seed=94)
        start = str(tmp_path / f"start_{n}.nbs")
        ds.save_snapshot(start, pos, vel, step=0, time=0.0)
        final = str(tmp_path / f"auto_{n}.nbs")
        out = run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--auto", "--final", final)
        assert f"force mode: {want}" in out
        p, v, _, _ = ds.load_snapshot(final)
        with nb.initialize(n, force_mode="auto") as s:
            assert s.force_mode == want.replace("-", "_")
            s.setParticlesPosition(pos)
            s.setParticlesVelocity(vel)
            s.step_n(3, 1e-3, 1e-3)
            want_p, want_v = s.download()
        assert np.array_equal(p, want_p) and np.array_equal(v, want_v), n
        final2 = str(tmp_path / f"auto_two_{n}.nbs")
        run_cli("--resume", start, "--steps", 3, "--dt", 1e-3, "--softening", 1e-3, "--auto", "--devices", "0,0", "--peer-copy",
                "--final", final2)
        p2, v2, _, _ = ds.load_snapshot(final2)
        with MultiGpuSystem(n, devices=[0, 0], force_mode="auto", transport="peer_copy") as m:
            assert m.force_mode == want.replace("-", "_")
            m.set_state(pos, vel)
            m.step_n(3, 1e-3, 1e-3)
            mp, mv = m.download()
        assert np.array_equal(p2, mp) and np.array_equal(v2, mv), n
