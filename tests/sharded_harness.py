"""TEST INFRASTRUCTURE (not part of the product package): rows sharded over several ranks with the exchange carried by
``torch.distributed`` -- the REHEARSAL harness.

The product's multi-GPU step lives behind the C ABI (``nbody_multi_*``, csrc/nbody_multi.hip, Python view
:class:`n_body_problem_amd.multi.MultiGpuSystem`): RCCL all-gather / ring inside the library, one process per GPU or
all GPUs in one process (``n_body_problem_amd.multi.sharded_system``).

What lives here is the same step spelled out in Python over ``torch.distributed`` with a host-staged backend (gloo),
for the two situations RCCL cannot serve: several ranks sharing ONE GPU (RCCL refuses duplicate devices) and ranks
without any GPU, where ``kernels_factory`` injects a CPU stand-in for the kernels (tests/_sharded_worker.py) so that
the sharding, the pair-once row/column/group data flow and both exchange schedules are checked under ``gloo`` with
world sizes 2 and 4.  Same geometry, same call order per rank, same bits as the library path:

* rank r integrates the contiguous rows ``[r*C, (r+1)*C)`` and holds a full replica of the positions;
* per step ONE exchange of the updated position slices, overlapped with the force kernel on the rank's OWN column
  chunk; ``exchange="ring"`` spells it out as P-1 send/recv hops, the force kernel of chunk (r-h) starting as hop h lands;
* chunk boundaries are multiples of ``split_len`` (whole split groups in the pair-once mode), every partial sum has one
  writer and a fixed order, so the state is bit-identical to the single-GPU run for any world size and exchange.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from n_body_problem_amd import system as _system


def shard_geometry(num_bodies: int, world_size: int, split_len: int):
    """(padded body count, rows per rank).  Rows per rank are a whole number of splits."""
    n_splits = max(1, -(-num_bodies // split_len))
    splits_per_rank = -(-n_splits // world_size)
    chunk = splits_per_rank * split_len
    return chunk * world_size, chunk


def pair_once_geometry(num_bodies: int, world_size: int, split_len: int):
    """(padded body count, rows per rank) of the pair-once mode: the partial sums are added in SYM_GROUPS groups of
    ceil(n_splits / SYM_GROUPS) splits and a rank owns whole groups, so the world size must divide SYM_GROUPS."""
    groups = _system.SYM_GROUPS
    if groups % world_size:
        raise ValueError(f"the pair-once mode shards over 1, 2, 4 or 8 ranks, not {world_size}")
    n_splits = max(1, -(-num_bodies // split_len))
    group_splits = -(-n_splits // groups)
    return groups * group_splits * split_len, (groups // world_size) * group_splits * split_len


def sym_rows_side(r: int, c: int, n_splits: int) -> bool:
    """nbody::sym_rows_side (csrc/nbody_kernels.h): is the tile of the split pair {r, c} computed with r's bodies as rows?"""
    d = (c - r) % n_splits
    if d == 0:
        return False
    if 2 * d != n_splits:
        return 2 * d < n_splits
    lo = min(r, c)
    return ((lo & 1) == 0) == (r == lo)


def ring_schedule(rank: int, world_size: int):
    """[(hop, chunk sent to rank+1, chunk received from rank-1)] for hops 1..P-1 of a ring all-gather."""
    return [(h, (rank - h + 1) % world_size, (rank - h) % world_size) for h in range(1, world_size)]


class ShardedNBodySystem:
    """The reference's step interface (see :mod:`n_body_problem_amd.system`) for one rank of a sharded run.

    ``kernels_factory(n_padded, row_lo, row_count, split_len)`` must return an object with the interface of
    :class:`NBodySystem` (``positions``, ``velocities``, ``forces``, ``forces_complement``, ``update``, ``sync``,
    ``energy``, ``momentum``).  The default -- and the only one the package ships -- is the HIP-backed ``NBodySystem``;
    tests inject a CPU stand-in to exercise the sharding and exchange logic under ``gloo``.
    """

    def __init__(self, num_bodies: int, group=None, device: Optional[int] = None, exchange: str = "allgather",
                 kernels_factory: Optional[Callable] = None, split_len: int = 0, integrator: str = "kick_drift",
                 force_mode: str = "one_sided"):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        if exchange not in ("allgather", "ring"):
            raise ValueError("exchange must be 'allgather' or 'ring'")
        if integrator not in ("kick_drift", "kdk"):
            raise ValueError("integrator must be 'kick_drift' or 'kdk'")
        if force_mode not in ("one_sided", "pair_once"):
            raise ValueError("force_mode must be 'one_sided' or 'pair_once'")
        self.exchange = exchange
        self.integrator = integrator
        self.force_mode = force_mode
        self._kdk_ready = False  # kdk: accelerations at the current positions are cached in the kernels object
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world_size = dist.get_world_size(group) if self.distributed else 1
        self.num_bodies = int(num_bodies)
        if kernels_factory is None:
            self.split_len = int(split_len) or (_system.pair_once_split_len(self.num_bodies) if force_mode == "pair_once" else
                                                _system.default_split_len(self.num_bodies))
        else:
            if not split_len:
                raise ValueError("a custom kernels_factory needs an explicit split_len")
            self.split_len = int(split_len)
        geometry = pair_once_geometry if force_mode == "pair_once" else shard_geometry
        self.n_padded, self.chunk = geometry(self.num_bodies, self.world_size, self.split_len)
        self.row_lo = self.rank * self.chunk
        if kernels_factory is None:
            dev = torch.cuda.current_device() if device is None else device
            self.kernels = _system.NBodySystem(self.n_padded, device=dev, row_lo=self.row_lo, row_count=self.chunk,
                                               split_len=self.split_len)
        else:
            self.kernels = kernels_factory(self.n_padded, self.row_lo, self.chunk, self.split_len)
        self._cp_send = None
        if force_mode == "pair_once":
            # each unordered pair once (nbody_symmetric.hip): besides the positions, the ranks exchange the sums of the
            # forces their rows put on everybody else's bodies -- SYM_GROUPS / P slices of (n_padded, 4) floats per rank
            # and step (16 MiB at N = 2^20, P = 8), one all-gather between the force kernels and the update
            self.kernels.set_force_mode("pair_once")
            if self.world_size > 1:
                self._cp_send = torch.empty_like(self.kernels.sym_own_slice())
        self.positions = self.kernels.positions      # full replica, (n_padded, 4)
        self.velocities = self.kernels.velocities    # own rows, (chunk, 4)
        self._on_gpu = bool(self.positions.is_cuda)
        backend = dist.get_backend(group) if self.distributed else "none"
        if backend == "nccl" and kernels_factory is None:
            raise ValueError("the RCCL exchange lives in the library: use sharded_system() / MultiGpuSystem.from_torch_distributed()")
        self._send = torch.empty_like(self.positions[:self.chunk])
        self._side_stream = torch.cuda.Stream(device=self.positions.device) if self._on_gpu else None
        self._pending = None   # allgather mode: work handle of the exchange in flight
        self._stale = False    # ring mode: remote chunks of the replica are one update behind

    # -- buffers ----------------------------------------------------------------------------------
    def _pad(self, data) -> np.ndarray:
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] != self.num_bodies:
            raise ValueError(f"expected {self.num_bodies} bodies, got {a.shape[0]}")
        out = np.zeros((self.n_padded, 4), dtype=np.float32)  # zero-mass bodies at the origin, as kernel.cu:265-277
        out[:self.num_bodies] = a
        return out

    def setParticlesPosition(self, data) -> None:
        self._refresh()
        self.positions.copy_(self._torch.from_numpy(self._pad(data)))
        self._kdk_ready = False

    def setParticlesVelocity(self, data) -> None:
        v = self._pad(data)[self.row_lo:self.row_lo + self.chunk]
        self.velocities.copy_(self._torch.from_numpy(np.ascontiguousarray(v)))

    set_particles_position = setParticlesPosition
    set_particles_velocity = setParticlesVelocity

    def set_particle_softening(self, eps) -> None:
        """Per-particle softening lengths of ALL ``num_bodies`` bodies, the same on every rank (the columns need every
        body's length; it is static, so there is nothing to exchange per step).  ``None`` switches it off."""
        if eps is None:
            self.kernels.set_particle_softening(None)
        else:
            e = np.ascontiguousarray(eps, dtype=np.float32).reshape(-1)
            if e.shape[0] != self.num_bodies:
                raise ValueError(f"expected {self.num_bodies} softening lengths, got {e.shape[0]}")
            padded = np.zeros(self.n_padded, dtype=np.float32)
            padded[:self.num_bodies] = e
            self.kernels.set_particle_softening(padded)
        self._kdk_ready = False

    def download(self):
        """Full (positions, velocities) of the real bodies on every rank (velocities are gathered)."""
        torch, dist = self._torch, self._dist
        self.sync()
        vel = self.velocities
        if self.world_size > 1:
            parts = [torch.empty_like(vel) for _ in range(self.world_size)]
            dist.all_gather(parts, vel.contiguous(), group=self.group)
            vel = torch.cat(parts)
        n = self.num_bodies
        return self.positions[:n].cpu().numpy(), vel[:n].cpu().numpy()

    # -- exchange -----------------------------------------------------------------------------------
    def _chunk(self, c: int):
        return self.positions[c * self.chunk:(c + 1) * self.chunk]

    def _drain(self) -> None:
        """allgather mode: make the current stream (for gloo: the host) wait for the exchange in flight."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def _start_allgather(self) -> None:
        self._send.copy_(self._chunk(self.rank))
        self._pending = self._dist.all_gather_into_tensor(self.positions, self._send, group=self.group, async_op=True)

    def _peer(self, r: int) -> int:
        r %= self.world_size
        return self._dist.get_global_rank(self.group, r) if self.group is not None else r

    def _ring_hop(self, send_c: int, recv_c: int) -> None:
        """One hop: chunk send_c goes to rank+1 while chunk recv_c arrives from rank-1, in place in the replica (blocking,
        staged through host memory when the replica is on a GPU)."""
        torch, dist = self._torch, self._dist
        nxt, prv = self._peer(self.rank + 1), self._peer(self.rank - 1)
        if self._on_gpu:
            torch.cuda.current_stream(self.positions.device).synchronize()
        send = self._chunk(send_c).cpu() if self._on_gpu else self._chunk(send_c)
        recv = torch.empty_like(send) if self._on_gpu else self._chunk(recv_c)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, nxt, group=self.group),
                                         dist.P2POp(dist.irecv, recv, prv, group=self.group)]):
            w.wait()
        if self._on_gpu:
            self._chunk(recv_c).copy_(recv)

    def _refresh(self) -> None:
        """Bring every chunk of the replica up to date without computing anything."""
        self._drain()
        if self._stale:
            for _, send_c, recv_c in ring_schedule(self.rank, self.world_size):
                self._ring_hop(send_c, recv_c)
            self._stale = False

    # -- the step -------------------------------------------------------------------------------------
    def _forces_all_columns(self, softening: float) -> None:
        """Partial sums of the own rows from every column: own chunk first (needs no remote data, so it runs beside
        the exchange in flight), the other chunks as they become current."""
        k = self.kernels
        lo = self.row_lo
        if self._stale:  # ring mode, remote chunks outstanding
            k.forces(lo, self.chunk, softening)              # own chunk: runs beside the first hops
            for _, send_c, recv_c in ring_schedule(self.rank, self.world_size):
                self._ring_hop(send_c, recv_c)
                k.forces(recv_c * self.chunk, self.chunk, softening)
            self._stale = False
            return
        torch = self._torch
        two_streams = self._on_gpu and self.world_size > 1
        if two_streams:
            cur = torch.cuda.current_stream(self.positions.device)
            self._side_stream.wait_stream(cur)           # uploads / the previous update are ordered before it
        k.forces(lo, self.chunk, softening)              # own chunk: runs beside the all-gather in flight
        if two_streams:
            # the other chunks on a second stream: their workgroups fill the CUs the first launch's tail
            # leaves idle (the launches write disjoint partial sums)
            with torch.cuda.stream(self._side_stream):
                self._drain()                            # this stream, not the host, waits for the exchange
                k.forces_complement(lo, self.chunk, softening)
            cur.wait_stream(self._side_stream)
        elif self.world_size > 1:
            self._drain()
            k.forces_complement(lo, self.chunk, softening)

    def _sum_forces(self) -> None:
        """Pair-once mode: the column-side sums of this rank's groups, for every body, gathered from every rank."""
        if self.force_mode != "pair_once":
            return
        k = self.kernels
        k.sym_reduce()
        if self.world_size > 1:
            self._cp_send.copy_(k.sym_own_slice())
            self._dist.all_gather_into_tensor(k.colparts, self._cp_send, group=self.group)

    def _exchange_own_rows(self) -> None:
        if self.world_size > 1:
            if self.exchange == "allgather":
                self._start_allgather()
            else:
                self._stale = True

    def step(self, dt: float = _system.TIME_TICK, softening: float = _system.SOFTENING_VERSION3,
             sync: bool = True) -> None:
        k = self.kernels
        if self.integrator == "kdk":
            # velocity Verlet: the drifted rows are exchanged BEFORE the forces; the own-chunk force kernel still
            # runs beside the exchange
            if not self._kdk_ready:
                self._forces_all_columns(softening)
                self._sum_forces()
                k.kdk_prepare()
                self._kdk_ready = True
            k.kdk_kick_drift(dt)
            self._exchange_own_rows()
            self._forces_all_columns(softening)
            self._sum_forces()
            k.kdk_kick(dt)
            if sync:
                self.sync()
            return
        self._forces_all_columns(softening)
        self._sum_forces()
        k.update(dt)
        self._exchange_own_rows()
        if sync:
            self.sync()

    def step_n(self, n: int, dt: float = _system.TIME_TICK, softening: float = _system.SOFTENING_VERSION3) -> None:
        for _ in range(int(n)):
            self.step(dt, softening, sync=False)
        self.sync()

    def sync(self) -> None:
        """Replica current on every rank and all device work complete."""
        self._refresh()
        self.kernels.sync()

    # -- diagnostics ------------------------------------------------------------------------------------
    def _allreduce(self, vals: np.ndarray) -> np.ndarray:
        if self.world_size == 1:
            return vals
        torch, dist = self._torch, self._dist
        t = torch.from_numpy(vals.copy()).to(self.positions.device)
        dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def energy(self, softening: float) -> np.ndarray:
        """[kinetic, potential, total] of the whole system (each rank's rows, summed over ranks)."""
        self.sync()
        return self._allreduce(np.asarray(self.kernels.energy(softening), dtype=np.float64))

    def momentum(self) -> np.ndarray:
        self.sync()
        return self._allreduce(np.asarray(self.kernels.momentum(), dtype=np.float64))

    def close(self) -> None:
        self._refresh()
        if hasattr(self.kernels, "close"):
            self.kernels.close()
