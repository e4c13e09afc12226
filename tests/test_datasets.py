"""CPU-only: the reference's dataset formats and the raw snapshot format (N3)."""
import csv
import os

import numpy as np
import pytest

from n_body_problem_amd import datasets as ds
from n_body_problem_amd import initial_conditions as ic

REF_DATA = "/root/reference/main_project/data"


def sample(n=37):
    pos, vel = ic.uniform_cube(n, seed=77, random_masses=True, speed=0.3)
    vel[:, 3] = np.linspace(0.01, 0.02, n, dtype=np.float32)   # per-particle eps travels in vel.w
    return pos, vel


def test_tipsy_round_trip_and_record_sizes(tmp_path):
    pos, vel = sample(37)
    f = str(tmp_path / "g.bin")
    ds.write_tipsy(f, pos, vel, ndark=5, time=1.5)
    assert os.path.getsize(f) == 32 + 5 * 36 + 32 * 44
    p, v = ds.read_tipsy(f)
    assert np.array_equal(p, pos) and np.array_equal(v, vel)
    # galaxy_20K.bin: 2500 dark + 17500 star = 860032 bytes
    assert 32 + 2500 * 36 + 17500 * 44 == 860032


def test_text_formats_round_trip(tmp_path):
    pos, vel = sample(20)
    f = str(tmp_path / "a.tab")
    ds.write_tab(f, pos, vel)
    open(f, "a").write("\n\n")                      # trailing blank lines must not become bodies (Q8)
    p, v = ds.read_tab(f)
    assert np.array_equal(p, pos) and np.array_equal(v[:, :3], vel[:, :3]) and np.all(v[:, 3] == 0)
    f = str(tmp_path / "a.dat")
    ds.write_dat(f, pos, vel)
    p, v = ds.read_dat(f)                            # z y x order on disk, mass forced to 1
    assert np.array_equal(p[:, :3], pos[:, :3]) and np.all(p[:, 3] == 1) and np.array_equal(v[:, :3], vel[:, :3])
    f = str(tmp_path / "a.snap")
    ds.write_snap(f, pos, vel, time=7.75)
    p, v = ds.read_snap(f)
    assert np.array_equal(p, pos) and np.array_equal(v, vel)
    assert np.array_equal(ds.read_any(f)[0], pos)
    with pytest.raises(ValueError):
        ds.read_any(str(tmp_path / "a.xyz"))


def test_snapshot_round_trip(tmp_path):
    pos, vel = sample(1000)
    f = str(tmp_path / "s.nbs")
    ds.save_snapshot(f, pos, vel, step=42, time=0.336)
    p, v, step, time = ds.load_snapshot(f)
    assert np.array_equal(p, pos) and np.array_equal(v, vel) and step == 42 and time == pytest.approx(0.336)
    open(f, "r+b").truncate(100)
    with pytest.raises(ValueError):
        ds.load_snapshot(f)


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's data files are not on this machine")
def test_reference_files_known_answers():
    """F4 of SURVEY.md 8c: galaxy_20K.bin parsed with the reference's structs equals its own CSV dump
    (unused_files/tool.cpp wrote galaxy_20k.csv from that file)."""
    pos, vel = ds.load_reference_dataset(0, REF_DATA)
    assert pos.shape == (20000, 4)
    rows = list(csv.reader(open(os.path.join(REF_DATA, "galaxy_20k.csv"))))[1:]
    assert len(rows) == 20000
    for i in (0, 1, 2, 2499, 2500, 19999):
        want = np.array([float(x) for x in rows[i][1:]])      # x,y,z,mass,vx,vy,vz,eps
        got = np.array([*pos[i, :3], pos[i, 3], *vel[i, :3], vel[i, 3]], dtype=np.float64)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-12), i   # the CSV keeps 6 significant digits
    p, v = ds.load_reference_dataset(3, REF_DATA)              # stars.dat: "z y x vz vy vx", mass 1
    first = [float(x) for x in open(os.path.join(REF_DATA, "stars.dat")).readline().split()]
    assert np.allclose(p[0, :3], first[2::-1], rtol=1e-6) and np.allclose(v[0, :3], first[5:2:-1], rtol=1e-6)
    assert p.shape[0] == 43802 and np.all(p[:, 3] == 1)      # 43837 lines, 35 records wrapped over two lines
    p, v = ds.load_reference_dataset(5, REF_DATA)              # k17hp.snap through the .snap parser
    assert p.shape == (10002, 4) and p[0, 3] == pytest.approx(2e-4)
    assert ic.padded_count(20000) == 20225                     # the count the reference hard-codes at kernel.cu:1130


def test_committed_galaxy_file_against_the_reference_csv_rows(golden_dir):
    """The reference's own input (load_data(0), kernel.cu:975-981) travels with the repo as a DATA fixture; nine rows of
    the CSV dump the reference keeps beside it (written by its unused_files/tool.cpp) are the known answer."""
    pos, vel = ds.read_tipsy(os.path.join(golden_dir, "galaxy_20K.bin"))
    assert pos.shape == (20000, 4) and vel.shape == (20000, 4)
    rows = list(csv.reader(open(os.path.join(golden_dir, "galaxy_20k_sample.csv"))))[1:]
    assert len(rows) == 9
    for r in rows:
        i = int(r[0])
        want = np.array([float(x) for x in r[2:]])              # x,y,z,mass,vx,vy,vz,eps after the CSV's own index
        got = np.array([*pos[i, :3], pos[i, 3], *vel[i, :3], vel[i, 3]], dtype=np.float64)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-12), i
    assert len(np.unique(pos[:, 3])) == 3                       # three mass species: halo, bulge, disk
    assert ic.padded_count(pos.shape[0]) == 20225               # kernel.cu:1130


def test_committed_k17hp_and_stars_fixtures(golden_dir):
    """The two further inputs of load_data that travel with the repo (tests/golden/README.md): k17hp.snap whole, the first
    8192 records of stars.dat.  Known answers read off the files' own text; where the reference is on this machine, the
    fixtures are its bytes / its tokens."""
    import hashlib
    p, v = ds.read_any(os.path.join(golden_dir, "k17hp.snap"))
    assert p.shape == (10002, 4) and np.all(p[:, 3] == np.float32(2e-4)) and np.isclose(p[:, 3].sum(dtype=np.float64), 2.0004, rtol=1e-6)
    tok = open(os.path.join(golden_dir, "k17hp.snap")).read().split()
    assert (int(tok[0]), int(tok[1])) == (10002, 3)
    first_pos = [float(x) for x in tok[3 + 10002:3 + 10002 + 3]]
    assert np.allclose(p[0, :3], first_pos, rtol=1e-6) and np.allclose(v[-1, 3], float(tok[-1]), rtol=1e-6)   # eps of the last body
    assert ic.padded_count(10002) == 10241
    p, v = ds.read_any(os.path.join(golden_dir, "k17c.snap"))
    assert p.shape == (32770, 4) and np.all(p[:, 3] == p[0, 3]) and np.isclose(p[:, 3].sum(dtype=np.float64), 2.0001, rtol=1e-4)
    tok = open(os.path.join(golden_dir, "k17c.snap")).read().split()
    assert (int(tok[0]), int(tok[1])) == (32770, 3) and np.allclose(p[0, :3], [float(x) for x in tok[3 + 32770:3 + 32770 + 3]], rtol=1e-6)
    p, v = ds.read_any(os.path.join(golden_dir, "stars_8192.dat"))
    first = [float(x) for x in open(os.path.join(golden_dir, "stars_8192.dat")).readline().split()]
    assert p.shape == (8192, 4) and np.all(p[:, 3] == 1) and np.all(v[:, 3] == 0)
    assert np.allclose(p[0, :3], first[2::-1], rtol=1e-6) and np.allclose(v[0, :3], first[5:2:-1], rtol=1e-6)   # z y x order on disk
    whole, wv = ds.read_any(os.path.join(golden_dir, "stars.dat"))        # load_data(3) at its real size (kernel.cu:996-1000)
    assert whole.shape == (43802, 4) and np.all(whole[:, 3] == 1) and np.all(wv[:, 3] == 0) and ic.padded_count(43802) == 44033
    assert np.array_equal(whole[:8192], p) and np.array_equal(wv[:8192], v)
    assert hashlib.sha256(open(os.path.join(golden_dir, "stars.dat"), "rb").read()).hexdigest() == \
        "836bb3b21b91c71343e49f612ff10c6e79091995d41d0f79df178dd4fd5b9556"
    if os.path.isdir(REF_DATA):
        sha = lambda f: hashlib.sha256(open(f, "rb").read()).hexdigest()
        assert sha(os.path.join(golden_dir, "stars.dat")) == sha(os.path.join(REF_DATA, "stars.dat"))
        assert sha(os.path.join(golden_dir, "k17hp.snap")) == sha(os.path.join(REF_DATA, "k17hp.snap"))
        assert sha(os.path.join(golden_dir, "k17c.snap")) == sha(os.path.join(REF_DATA, "k17c.snap"))
        full, _ = ds.read_dat(os.path.join(REF_DATA, "stars.dat"))
        assert np.array_equal(full[:8192], p)
