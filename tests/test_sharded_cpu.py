"""CPU-only: the multi-GPU host logic (sharding geometry, exchange schedule, overlap ordering, padding)
under torch.distributed/gloo with world_size 2, with a CPU stand-in for the HIP kernels.  What is
checked is the part that cannot be seen on a one-GPU box: that P ranks reproduce the one-rank state
bit for bit in both exchange modes."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_state_error
from _sharded_worker import run_rank


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_geometry_and_ring_schedule():
    from sharded_harness import shard_geometry, ring_schedule
    assert shard_geometry(1 << 20, 8, 65536) == (1 << 20, 131072)
    assert shard_geometry(1 << 20, 1, 65536) == (1 << 20, 1 << 20)
    assert shard_geometry(1000, 2, 256) == (1024, 512)
    assert shard_geometry(1000, 3, 256) == (1536, 512)      # 4 splits over 3 ranks: 2 each, zero-mass padding
    assert shard_geometry(1, 8, 256) == (2048, 256)
    for P in (2, 3, 8):
        have = {r: {r} for r in range(P)}
        for h in range(1, P):
            sends = {r: ring_schedule(r, P)[h - 1] for r in range(P)}
            for r in range(P):
                _, send_c, recv_c = sends[r]
                assert send_c in have[r]                                 # a rank only forwards what it holds
                assert sends[(r - 1) % P][1] == recv_c                   # and receives what its neighbour sends
            for r in range(P):
                have[r].add(sends[r][2])
        assert all(have[r] == set(range(P)) for r in range(P))


@pytest.mark.parametrize("exchange", ["allgather", "ring"])
def test_two_ranks_reproduce_one_rank_bit_for_bit(tmp_path, oracle_mod, exchange):
    import torch.multiprocessing as mp
    n, split_len, steps = 1000, 256, 3
    out = str(tmp_path)
    run_rank(0, 1, 0, exchange, n, split_len, steps, out)
    mp.spawn(run_rank, args=(2, free_port(), exchange, n, split_len, steps, out), nprocs=2, join=True)
    one = np.load(os.path.join(out, f"w1_{exchange}_r0.npz"))
    r0 = np.load(os.path.join(out, f"w2_{exchange}_r0.npz"))
    r1 = np.load(os.path.join(out, f"w2_{exchange}_r1.npz"))
    assert int(r0["n_padded"]) == 1024 and int(r0["chunk"]) == 512
    # every rank ends with the same, complete state, identical to the one-rank run
    for r in (r0, r1):
        assert np.array_equal(r["p"], one["p"]) and np.array_equal(r["v"], one["v"])
        assert np.allclose(r["e1"], one["e1"], rtol=1e-12) and np.allclose(r["mom"], one["mom"], atol=1e-15)
    # own chunk is integrated first, the remote one after the exchange
    assert tuple(r0["calls"][-2]) == (0, 512) and tuple(r1["calls"][-2]) == (512, 512)
    if exchange == "allgather":
        assert "complement" in set(r0["kinds"].tolist())
    # and the sharded state is the oracle's state up to summation order
    from n_body_problem_amd import initial_conditions as ic
    pos, vel = ic.plummer(n, seed=1234)
    pr, vr = oracle_mod.step_f32(pos, vel, 1e-3, 1e-2, nsteps=steps)
    assert rel_state_error(one["p"], pr) < 1e-6 and rel_state_error(one["v"], vr) < 1e-6
    assert abs(one["e1"][2] - one["e0"][2]) / abs(one["e0"][2]) < 1e-3
    assert np.allclose(one["e0"], oracle_mod.energy(pos, vel, 1e-2), rtol=1e-9)


def test_two_ranks_kdk_reproduce_one_rank(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    n, split_len, steps = 1000, 256, 3
    out = str(tmp_path)
    run_rank(0, 1, 0, "allgather", n, split_len, steps, out, "kdk")
    mp.spawn(run_rank, args=(2, free_port(), "allgather", n, split_len, steps, out, "kdk"), nprocs=2, join=True)
    one = np.load(os.path.join(out, "w1_allgather_kdk_r0.npz"))
    for r in range(2):
        g = np.load(os.path.join(out, f"w2_allgather_kdk_r{r}.npz"))
        assert np.array_equal(g["p"], one["p"]) and np.array_equal(g["v"], one["v"])
    from n_body_problem_amd import initial_conditions as ic
    pos, vel = ic.plummer(n, seed=1234)
    pr, vr = oracle_mod.step_kdk_f32(pos, vel, 1e-3, 1e-2, nsteps=steps)
    assert rel_state_error(one["p"], pr) < 1e-6 and rel_state_error(one["v"], vr) < 1e-6


def test_pair_once_geometry_and_tile_orientation():
    from sharded_harness import pair_once_geometry, sym_rows_side
    assert pair_once_geometry(1 << 20, 8, 2048) == (1 << 20, 131072)      # 512 splits, 8 groups of 64
    assert pair_once_geometry(1 << 20, 1, 2048) == (1 << 20, 1 << 20)
    assert pair_once_geometry(1000, 2, 256) == (2048, 1024)               # 4 splits -> 8 groups of 1 (4 of padding)
    assert pair_once_geometry(20000, 4, 256) == (20480, 5120)             # 79 splits -> 8 groups of 10
    with pytest.raises(ValueError):
        pair_once_geometry(1000, 3, 256)
    for S in (1, 2, 5, 8, 16, 79):
        rows = [0] * S
        for r in range(S):
            assert not sym_rows_side(r, r, S)
            for c in range(r + 1, S):
                assert sym_rows_side(r, c, S) != sym_rows_side(c, r, S)   # every unordered pair exactly once
                rows[r if sym_rows_side(r, c, S) else c] += 1
        assert max(rows) - min(rows) <= 1                                 # and the same work for every split


@pytest.mark.parametrize("world,exchange,integrator", [(2, "allgather", "kick_drift"), (2, "ring", "kick_drift"),
                                                       (4, "allgather", "kdk")])
def test_pair_once_ranks_reproduce_one_rank_bit_for_bit(tmp_path, oracle_mod, world, exchange, integrator):
    """The pair-once data flow (row sums, column sums, per-group reduction, one all-gather, fixed summation order) under
    gloo: 2 and 4 ranks end with the bits of the one-rank run."""
    import torch.multiprocessing as mp
    n, split_len, steps = 1000, 256, 2
    out = str(tmp_path)
    run_rank(0, 1, 0, exchange, n, split_len, steps, out, integrator, "pair_once")
    mp.spawn(run_rank, args=(world, free_port(), exchange, n, split_len, steps, out, integrator, "pair_once"),
             nprocs=world, join=True)
    tag = (exchange if integrator == "kick_drift" else exchange + "_" + integrator) + "_pair_once"
    one = np.load(os.path.join(out, f"w1_{tag}_r0.npz"))
    assert int(one["n_padded"]) == 2048
    for r in range(world):
        g = np.load(os.path.join(out, f"w{world}_{tag}_r{r}.npz"))
        assert int(g["chunk"]) == 2048 // world
        assert np.array_equal(g["p"], one["p"]) and np.array_equal(g["v"], one["v"])
    from n_body_problem_amd import initial_conditions as ic
    pos, vel = ic.plummer(n, seed=1234)
    step = oracle_mod.step_f32 if integrator == "kick_drift" else oracle_mod.step_kdk_f32
    pr, vr = step(pos, vel, 1e-3, 1e-2, nsteps=steps)
    assert rel_state_error(one["p"], pr) < 1e-6 and rel_state_error(one["v"], vr) < 1e-6
