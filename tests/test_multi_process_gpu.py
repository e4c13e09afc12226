"""ONE RANK PER PROCESS -- the model the benchmark's ranks run under torchrun on a multi-GPU node (nbody_multi_create_rank,
include/nbody.h) -- executed by 2-4 processes that share cuda:0.  Real RCCL refuses two ranks on one device, so these
processes load a second build of the library in which the RCCL entry points are a test double (tests/fake_rccl; the product
library always links the real librccl).  Since round 4 the double keeps RCCL's STREAM semantics -- calls return at once, the
bytes move on the caller's stream behind what the caller ordered in front of them, other streams see the result only through
the events the library records -- so a missing hipStreamWaitEvent in the library's exchange turns these tests red
(tools/edge_mutations.py, profiles/r04_edge_mutations.txt), and its communicators are non-blocking like the product's.  What this covers that the
single-process peer-copy tests cannot: every place where the library must tell a LOCAL index from a GLOBAL rank, the order
of the collective calls each rank makes, the id hand-over, the collective download / diagnostics, and what a rank reports
when a peer is gone.  The results must equal the single-process run of the same configuration bit for bit."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT, EPS = 1e-3, 1e-3


@pytest.fixture(scope="module")
def fake_library():
    sys.path.insert(0, os.path.join(ROOT, "tests", "fake_rccl"))
    import build_fake_rccl as fake_build
    return fake_build.build()


def run_ranks(fake_library, world, cfg, expect_failure=False, extra_env=None):
    work = tempfile.mkdtemp(prefix="nbody_ranks_")
    env = dict(os.environ, NBODY_AMD_LIBRARY=fake_library, FAKE_RCCL_SLOT_MB="32", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_multi_rank_worker.py"), str(r), str(world),
                               json.dumps(cfg), work], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung:\n" + "\n".join(logs))
        logs.append(out)
    if not expect_failure:
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return work, [p.returncode for p in procs], logs


def single_process(cfg, world):
    """The same configuration with every rank in THIS process (peer copies): the established bit-exact twin of one context."""
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    n = cfg["n"]
    pos, vel = nb.plummer(n, seed=cfg.get("seed", 77))
    rng = np.random.default_rng(5)
    if cfg.get("random_masses"):
        pos[:, 3] *= rng.uniform(0.5, 2.0, n).astype(np.float32)
    with MultiGpuSystem(n, devices=[0] * world, force_mode=cfg["force_mode"], integrator=cfg["integrator"],
                        exchange=cfg["exchange"], transport="peer_copy", split_len=cfg.get("split_len", 0),
                        body_order=cfg["body_order"]) as m:
        m.set_state(pos, vel)
        if cfg.get("pps"):
            m.set_particle_softening((rng.random(n) * 0.02).astype(np.float32))
        if cfg.get("reorder_every"):
            m.set_reorder_period(cfg["reorder_every"])
        m.step(cfg["dt"], cfg["eps"])
        m.step_n(cfg["steps"] - 1, cfg["dt"], cfg["eps"])
        if cfg.get("reorder"):
            m.reorder()
            m.step_n(2, cfg["dt"], cfg["eps"])
        p, v = m.download()
        return p, v, m.energy(cfg["eps"]), m.momentum()


CASES = [
    (2, dict(n=5000, force_mode="one_sided", integrator="kick_drift", exchange="allgather", body_order="given", steps=3)),
    (3, dict(n=7001, force_mode="one_sided", integrator="kdk", exchange="ring", body_order="given", steps=3)),
    (2, dict(n=40000, force_mode="pair_once", integrator="kdk", exchange="ring", body_order="morton", steps=3, split_len=512,
             reorder=True)),
    (4, dict(n=30000, force_mode="pair_once", integrator="kick_drift", exchange="allgather", body_order="morton", steps=4,
             split_len=1024, pps=True, random_masses=True, reorder_every=2)),
    (4, dict(n=20000, force_mode="auto", integrator="kick_drift", exchange="ring", body_order="morton", steps=3)),
]


@pytest.mark.parametrize("world,cfg", CASES)
def test_one_rank_per_process_equals_the_single_process_run(fake_library, world, cfg):
    cfg = dict(cfg, dt=DT, eps=EPS)
    work, _, logs = run_ranks(fake_library, world, cfg)
    want_p, want_v, want_e, want_mom = single_process(cfg, world)
    for r in range(world):
        meta = json.load(open(os.path.join(work, f"rank{r}.json")))
        assert meta["rccl_ranks"] == world and meta["world"] == world and meta["local_ranks"] == 1 and meta["rank"] == r
        assert meta["replicas_identical"]
        got = np.load(os.path.join(work, f"rank{r}.npz"))
        assert np.array_equal(got["p"], want_p) and np.array_equal(got["v"], want_v), (r, "\n".join(logs))
        assert np.allclose(got["e"], want_e, rtol=1e-12, atol=0) and np.allclose(got["mom"], want_mom, rtol=1e-9, atol=1e-12)
        tm = meta["timing"]
        assert tm["steps"] >= cfg["steps"] and tm["force_launches"] > 0 and tm["pos_exchanges"] > 0


@pytest.mark.parametrize("world,cfg", [
    (2, dict(n=6000, force_mode="one_sided", integrator="kdk", exchange="ring", body_order="given", steps=3)),
    (4, dict(n=30000, force_mode="pair_once", integrator="kick_drift", exchange="allgather", body_order="morton", steps=4,
             split_len=512, pps=True, random_masses=True, reorder_every=2)),
    (8, dict(n=50000, force_mode="pair_once", integrator="kdk", exchange="ring", body_order="morton", steps=3, split_len=256,
             reorder=True))])
def test_all_ranks_in_one_process_over_the_rccl_calls(fake_library, world, cfg):
    """The other way the library uses RCCL: ONE process, one communicator per device from ncclCommInitAll, the calls of all
    ranks inside one group (nbody_multi_create with NBODY_TRANSPORT_RCCL: host/nbody_run --devices ..., MultiGpuSystem(devices=
    [...])).  All ranks on cuda:0 over the test double; equal to the peer-copy transport bit for bit."""
    cfg = dict(cfg, dt=DT, eps=EPS)
    work = tempfile.mkdtemp(prefix="nbody_ranks_")
    env = dict(os.environ, NBODY_AMD_LIBRARY=fake_library, FAKE_RCCL_SLOT_MB="32")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_multi_rank_worker.py"), "all_local", str(world),
                          json.dumps(cfg), work], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    meta = json.load(open(os.path.join(work, "rank0.json")))
    assert meta["rccl_ranks"] == world and meta["local_ranks"] == world and meta["replicas_identical"]
    want_p, want_v, want_e, want_mom = single_process(cfg, world)
    got = np.load(os.path.join(work, "rank0.npz"))
    assert np.array_equal(got["p"], want_p) and np.array_equal(got["v"], want_v)
    assert np.allclose(got["e"], want_e, rtol=1e-12, atol=0) and np.allclose(got["mom"], want_mom, rtol=1e-9, atol=1e-12)


def test_eight_ranks_as_eight_threads_of_one_process(fake_library):
    """World size 8 (one row group per rank in the pair-once mode: the driver's largest run) -- as threads, because the box
    allows six processes on its GPU.  Each thread owns one nbody_multi in the one-rank-per-process model."""
    cfg = dict(n=50000, force_mode="pair_once", integrator="kick_drift", exchange="allgather", body_order="morton", steps=3,
               split_len=512, reorder_every=2, dt=DT, eps=EPS)
    work = tempfile.mkdtemp(prefix="nbody_ranks_")
    env = dict(os.environ, NBODY_AMD_LIBRARY=fake_library, FAKE_RCCL_SLOT_MB="32")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_multi_rank_worker.py"), "threads", "8", json.dumps(cfg), work],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    want_p, want_v, _, _ = single_process(cfg, 8)
    for r in range(8):
        meta = json.load(open(os.path.join(work, f"rank{r}.json")))
        assert meta["rccl_ranks"] == 8 and meta["rank"] == r and meta["replicas_identical"]
        got = np.load(os.path.join(work, f"rank{r}.npz"))
        assert np.array_equal(got["p"], want_p) and np.array_equal(got["v"], want_v), r


def test_a_rank_that_never_steps_is_reported_by_the_others(fake_library):
    """Rank 1 creates its communicator and leaves.  The survivors must come back with an error (the library's wording, the
    RCCL call it was in), not hang: bench.py turns that into its JSON error line."""
    cfg = dict(n=4096, force_mode="one_sided", integrator="kick_drift", exchange="allgather", body_order="given", steps=3,
               dt=DT, eps=EPS, die=1, timeout=10.0)
    work, codes, logs = run_ranks(fake_library, 2, cfg, expect_failure=True, extra_env={"FAKE_RCCL_TIMEOUT_S": "5"})
    assert codes[1] == 0 and codes[0] == 3, "\n".join(logs)
    meta = json.load(open(os.path.join(work, "rank0.json")))
    assert "error" in meta and "RCCL" in meta["error"]


def test_bench_ranks_under_torchrun_with_the_rendezvous_on_gloo(fake_library):
    """bench.py's product branch (one process per rank, library-owned exchange, per-rank breakdown, max over ranks) the way
    the driver starts it, with two ranks on cuda:0."""
    env = dict(os.environ, NBODY_AMD_LIBRARY=fake_library, NBODY_RENDEZVOUS="gloo", FAKE_RCCL_SLOT_MB="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--bodies",
           "65536", "--single-device", "--no-extra-legs"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [json.loads(x) for x in res.stdout.splitlines() if x.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["value"] > 0
    assert line["config"]["exchange_owner"].startswith("library")
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert line["sanity"]["position_replicas_identical_on_all_ranks"] and abs(line["sanity"]["dE_over_E0"]) < 1e-4


def test_long_run_tool_under_torchrun(fake_library):
    """tools/run_sharded.py (BASELINE config 5's command, one process per GPU) with two ranks on cuda:0: energy reports, the
    layout refreshed on the device, the final JSON summary."""
    env = dict(os.environ, NBODY_AMD_LIBRARY=fake_library, NBODY_RENDEZVOUS="gloo", FAKE_RCCL_SLOT_MB="32")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "tools", "run_sharded.py"), "--bodies", "32768", "--steps", "20",
           "--energy-every", "10", "--body-order", "morton", "--reorder-every", "5", "--single-device"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [json.loads(x) for x in res.stdout.splitlines() if x.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 20 and lines[0]["max_abs_dE_over_E0"] < 1e-4
    assert sum(x.startswith("step ") for x in res.stdout.splitlines()) == 2


@pytest.mark.parametrize("world,flags", [(2, ["--pair-once", "--kdk"]), (4, ["--auto", "--morton", "--reorder-every", "2", "--ring"])])
def test_cpp_host_one_process_per_gpu(fake_library, tmp_path, world, flags):
    """host/nbody_run with --rank / --world / --id-file (one process per GPU: the thin C++ host under mpirun or a job script),
    the ranks sharing cuda:0 over the test double: energy reports and snapshots from rank 0, and the final state equal to the
    single-process run (--devices 0,0,... --peer-copy) byte for byte."""
    from n_body_problem_amd import build as product
    import build_fake_rccl as fake_build
    exe = fake_build.build_host()
    common = ["--plummer", "40000", "--steps", "6", "--dt", "1e-3", "--softening", "1e-3", "--energy-every", "3", *flags]
    id_file = tmp_path / "comm.id"
    env = dict(os.environ, FAKE_RCCL_SLOT_MB="32")
    procs = [subprocess.Popen([exe, *common, "--rank", str(r), "--world", str(world), "--device", "0", "--id-file", str(id_file),
                               "--final", str(tmp_path / "ranks.nbs")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung:\n" + "\n".join(outs))
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    assert "dE/E0" in outs[0] and "replicas identical: yes" in outs[0] and f"RCCL communicator of {world}" in outs[0]
    assert all("dE/E0" not in o for o in outs[1:])                  # rank 0 reports
    assert not id_file.exists()                                     # and removes the id file
    one = subprocess.run([product.build_host(), *common, "--devices", ",".join(["0"] * world), "--peer-copy", "--final",
                          str(tmp_path / "one.nbs")], capture_output=True, text=True, timeout=240)
    assert one.returncode == 0, one.stderr
    assert (tmp_path / "ranks.nbs").read_bytes() == (tmp_path / "one.nbs").read_bytes()
    energies = lambda text: [ln.split("E =")[1].split()[0] for ln in text.splitlines() if ln.startswith("step ")]  # noqa: E731
    assert energies(outs[0]) == energies(one.stdout)


def test_real_rccl_follows_the_non_blocking_protocol_the_library_uses():
    """tests/rccl_probe against the REAL librccl with the one rank this box has: ncclCommInitRankConfig(blocking = 0), the poll
    of ncclCommGetAsyncError, a group of all-gather + send / recv settled before an event is recorded behind it; and a world
    of two with the peer absent, the creation in a helper thread with a limit as the library does it: the caller is back in
    time whether this RCCL honours blocking = 0 for the creation or sits in its bootstrap inside the call (RCCL 2.27.7 of
    ROCm 7.2 does the latter -- which is why the library does not rely on the flag)."""
    import build_fake_rccl as fake_build
    exe = fake_build.build_probe()
    one = subprocess.run([exe, "one"], capture_output=True, text=True, timeout=300)
    print(one.stdout.strip())
    assert one.returncode == 0 and "0 wrong words" in one.stdout, one.stdout + one.stderr
    absent = subprocess.run([exe, "absent"], capture_output=True, text=True, timeout=120)
    print(absent.stdout.strip())
    assert absent.returncode == 0 and ("has not returned" in absent.stdout or "honoured" in absent.stdout), absent.stdout + absent.stderr


_LEAVES_AFTER_THE_RENDEZVOUS = """
import os, time
import torch.distributed as dist
dist.init_process_group("gloo")
box = [None]
dist.broadcast_object_list(box, src=0)      # the RCCL id arrives ...
time.sleep(1.0)
os._exit(0)                                  # ... and this rank never creates its communicator
"""


@pytest.mark.parametrize("library", ["test double", "real RCCL"])
def test_bench_reports_a_rank_that_never_creates_its_communicator(fake_library, library):
    """The first contact: a job whose rank 1 is gone before it creates its communicator.  The communicators are non-blocking,
    so rank 0's nbody_multi_create_rank polls the creation under --exchange-timeout and comes back with NBODY_ERR_DEVICE, and
    bench.py prints ONE JSON line with "error" and leaves with a non-zero code -- instead of sitting in RCCL's bootstrap until
    the driver's limit.  With the test double and with the real librccl (whose bootstrap really waits for the peer)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, NBODY_RENDEZVOUS="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), FAKE_RCCL_TIMEOUT_S="40")
    if library == "test double":
        env["NBODY_AMD_LIBRARY"] = fake_library
    else:
        env.pop("NBODY_AMD_LIBRARY", None)
    peer = subprocess.Popen([sys.executable, "-c", _LEAVES_AFTER_THE_RENDEZVOUS], env=dict(env, RANK="1", LOCAL_RANK="1"),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    import time
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "8192", "--steps", "2",
                          "--warmup", "1", "--single-device", "--no-extra-legs", "--exchange-timeout", "4"], env=env,
                         capture_output=True, text=True, timeout=240)
    took = time.time() - t0
    peer.communicate(timeout=60)
    lines = [json.loads(x) for x in res.stdout.splitlines() if x.startswith("{")]
    assert res.returncode != 0 and len(lines) == 1 and "error" in lines[0], res.stdout[-2000:] + res.stderr[-3000:]
    assert "creating the RCCL communicators" in lines[0]["error"] and "timed out" in lines[0]["error"], lines[0]["error"]
    assert took < 120, took                                  # interpreter start + the 4 s timeout, not a hang
