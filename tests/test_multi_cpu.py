"""CPU-only: the host-visible logic of the library-owned multi-GPU step (include/nbody.h, nbody_multi_*) through the C ABI:
sharding geometry and ring schedule (pure functions of the library), the loud failure without a device, and -- under
torch.distributed/gloo with world size 2 -- the one-process-per-GPU plumbing: rank 0's RCCL id reaches every rank and
every rank's nbody_multi_create_rank fails with NBODY_ERR_NO_DEVICE instead of falling back to anything."""
import os
import socket

import numpy as np
import pytest


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_geometry_matches_the_rehearsal_harness_and_keeps_whole_groups():
    import sharded_harness as sharded
    from n_body_problem_amd import multi, system
    for n in (1, 1000, 20000, 65536, 1 << 20, (1 << 22) + 5):
        for world in (1, 2, 3, 4, 8):
            split = system.default_split_len(n)
            assert multi.geometry(n, world) == (*sharded.shard_geometry(n, world, split), split)
            if 8 % world == 0:
                ps = system.pair_once_split_len(n)
                padded, chunk, got = multi.geometry(n, world, "pair_once")
                assert (padded, chunk) == sharded.pair_once_geometry(n, world, ps) and got == ps
                group = padded // 8                       # a rank owns whole groups of ceil(n_splits / 8) splits
                assert chunk % group == 0 and group % ps == 0 and chunk * world == padded >= n
    assert multi.geometry(1 << 20, 8, "pair_once") == (1 << 20, 131072, 2048)      # BASELINE configs[3]
    assert multi.geometry(1 << 22, 8, "pair_once") == (1 << 22, 524288, 2048)      # BASELINE configs[4]
    assert multi.geometry(1000, 2, split_len=256) == (1024, 512, 256)
    with pytest.raises(ValueError):
        multi.geometry(1 << 20, 3, "pair_once")           # 3 does not divide the 8 summation groups
    with pytest.raises(ValueError):
        multi.geometry(1000, 2, split_len=100)            # not a multiple of the 256-body tile


def test_ring_schedule_delivers_every_chunk_exactly_once():
    import sharded_harness as sharded
    from n_body_problem_amd import multi
    for P in (1, 2, 3, 4, 8):
        assert all(multi.ring_schedule(r, P) == sharded.ring_schedule(r, P) for r in range(P))
        have = {r: [r] for r in range(P)}
        for h in range(1, P):
            hop = {r: multi.ring_schedule(r, P)[h - 1] for r in range(P)}
            for r in range(P):
                _, send_c, recv_c = hop[r]
                assert send_c in have[r]                            # a rank only forwards what it already holds
                assert hop[(r - 1) % P][1] == recv_c                # and receives what its left neighbour sends
            for r in range(P):
                have[r].append(hop[r][2])
        assert all(sorted(have[r]) == list(range(P)) for r in range(P))      # every chunk once, none twice
    with pytest.raises(ValueError):
        multi.ring_schedule(2, 2)


def test_no_device_is_a_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    from n_body_problem_amd import multi, _lib
    for kw in ({}, {"transport": "peer_copy"}, {"force_mode": "pair_once", "exchange": "ring"}):
        with pytest.raises(_lib.NBodyError) as e:
            multi.MultiGpuSystem(1000, devices=[0], **kw)
        assert e.value.status == _lib.NBODY_ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_bad_configurations_are_rejected_before_any_device_work():
    import ctypes
    from n_body_problem_amd import _lib
    lib = _lib.load()
    m = ctypes.c_void_p(None)
    dev = (ctypes.c_int * 3)(0, 1, 2)
    bad = [(_lib.MultiConfig(1000, 0, 1, 0, 0, 0), 3),      # pair-once over 3 ranks
           (_lib.MultiConfig(1000, 100, 0, 0, 0, 0), 2),    # split_len not a multiple of 256
           (_lib.MultiConfig(1000, 0, 7, 0, 0, 0), 2),      # unknown force mode
           (_lib.MultiConfig(1000, 0, 0, 0, 5, 0), 2),      # unknown exchange
           (_lib.MultiConfig(1000, 0, 0, 0, 0, 0, 2, 0), 2),  # unknown body order
           (_lib.MultiConfig(-1, 0, 0, 0, 0, 0), 2)]
    for cfg, n in bad:
        assert lib.nbody_multi_create(ctypes.byref(m), ctypes.byref(cfg), dev, n) == _lib.NBODY_ERR_INVALID
        assert not m.value and lib.nbody_multi_last_error(None)
    ident = ctypes.create_string_buffer(128)
    cfg = _lib.MultiConfig(1000, 0, 0, 0, 0, 1)              # peer copies cannot cross processes
    assert lib.nbody_multi_create_rank(ctypes.byref(m), ctypes.byref(cfg), 0, 5, 2, ident) == _lib.NBODY_ERR_INVALID
    assert lib.nbody_multi_create_rank(ctypes.byref(m), ctypes.byref(cfg), 0, 0, 2, None) == _lib.NBODY_ERR_INVALID
    assert lib.nbody_multi_step(None, 0.1, 0.1) == _lib.NBODY_ERR_INVALID and lib.nbody_multi_destroy(None) == 0
    assert lib.nbody_multi_reorder(None) == _lib.NBODY_ERR_INVALID and lib.nbody_multi_order(None, None) == _lib.NBODY_ERR_INVALID
    assert lib.nbody_multi_set_reorder_period(None, 5) == _lib.NBODY_ERR_INVALID


def _rank_without_gpu(rank, world, port, out_dir):
    import hashlib
    import torch
    import torch.distributed as dist
    from n_body_problem_amd import multi, _lib
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        seen = {}
        real = multi.unique_id

        def spy():
            seen["made"] = real()
            return seen["made"]
        multi.unique_id = spy
        status, text = None, ""
        try:
            multi.MultiGpuSystem.from_torch_distributed(4096, device=0, force_mode="pair_once")
        except _lib.NBodyError as e:
            status, text = e.status, str(e)
        ids = [None] * world
        # what create_rank was handed is not observable from outside: re-run the broadcast the constructor performs
        box = [seen.get("made")]
        dist.broadcast_object_list(box, src=0)
        dist.all_gather_object(ids, hashlib.sha256(box[0]).hexdigest())
        np.savez(os.path.join(out_dir, f"multi_cpu_r{rank}.npz"), status=-99 if status is None else status, text=text,
                 made=("made" in seen), same=len(set(ids)) == 1, gpu=torch.cuda.is_available())
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo_carries_the_rccl_id_and_every_rank_fails_loudly(tmp_path):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    from n_body_problem_amd import _lib
    mp.spawn(_rank_without_gpu, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), f"multi_cpu_r{r}.npz"))
        assert bool(g["made"]) == (r == 0)                  # only rank 0 asks RCCL for an id
        assert bool(g["same"])                              # and every rank ends up with those 128 bytes
        assert int(g["status"]) == _lib.NBODY_ERR_NO_DEVICE and "no CPU path" in str(g["text"])
