#!/usr/bin/env python3
"""Probe: does RCCL accept two ranks (two processes) on ONE device?  If it did, the multi-process leg of the library-owned
exchange could be tested on a one-GPU box; it is expected to refuse ("duplicate GPU"), which is why the one-GPU tests use
NBODY_TRANSPORT_PEER_COPY.  Prints what happened; never hangs longer than the library's exchange timeout set below."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, world, port):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["NBODY_EXCHANGE_TIMEOUT_S"] = "60"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import n_body_problem_amd as nb
    from n_body_problem_amd.multi import MultiGpuSystem
    try:
        m = MultiGpuSystem.from_torch_distributed(8192, device=0)
        pos, vel = nb.plummer(8192, seed=3)
        m.set_state(pos, vel)
        m.step_n(2, 1e-3, 1e-3)
        p, v = m.download()
        print(f"rank {rank}: RCCL accepted two ranks on one device; info {m.info()} replicas identical {m.replicas_identical()} "
              f"checksum {float(np.abs(p).sum()):.6f}", flush=True)
        m.close()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(rank_main, args=(2, port), nprocs=2, join=True)
