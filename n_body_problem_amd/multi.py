"""Rows sharded over the GPUs of one node with the exchange INSIDE the library (``nbody_multi_*`` of include/nbody.h).

The reference is single-GPU (nothing to mirror); this is the reference's step interface --
``initialize`` / ``setParticlesPosition`` / ``setParticlesVelocity`` / ``step`` (main_project/kernel.cu:130-188,
1225-1242) -- for a body set dealt to P GPUs.  Everything per step happens behind the C ABI: the force launches, the
RCCL all-gather (or ring) of the updated rows overlapped with the own-chunk launch, the pair-once column-sum
all-gather, the update, and the polling of RCCL's asynchronous error state.  Python only creates the object and, in
the one-process-per-GPU model, carries the 128-byte RCCL id from rank 0 to the other ranks over ``torch.distributed``.

Two process models:

* ``MultiGpuSystem(n, devices=[0, 1, ...])`` -- every rank in this process (``nbody_multi_create``);
* ``MultiGpuSystem.from_torch_distributed(n, device)`` -- one rank per process, the launch model of ``torchrun``
  (``nbody_multi_create_rank``).

There is no CPU or PyTorch fallback: the constructors raise without the built library or a gfx950 device.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import system as _system
from ._lib import NBodyError

FORCE_MODES = {"one_sided": 0, "pair_once": 1, "symmetric": 1, "auto": 2}
INTEGRATORS = {"kick_drift": 0, "kdk": 1}
EXCHANGES = {"allgather": 0, "ring": 1}
TRANSPORTS = {"rccl": 0, "peer_copy": 1}
UNIQUE_ID_BYTES = 128


def _check(status: int, handle=None) -> None:
    if status != _lib.NBODY_OK:
        lib = _lib.load()
        msg = lib.nbody_multi_last_error(handle) or b""
        raise NBodyError(status, msg.decode("utf-8", "replace") or lib.nbody_status_string(status).decode())


def geometry(num_bodies: int, world_size: int, force_mode: str = "one_sided", split_len: int = 0) -> Tuple[int, int, int]:
    """(padded body count, rows per rank, split length) -- ``nbody_multi_geometry``; needs no device."""
    padded, chunk, split = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    rc = _lib.load().nbody_multi_geometry(int(num_bodies), int(world_size), FORCE_MODES[force_mode], int(split_len),
                                          ctypes.byref(padded), ctypes.byref(chunk), ctypes.byref(split))
    if rc != _lib.NBODY_OK:
        raise ValueError(f"no sharding of {num_bodies} bodies over {world_size} ranks in the {force_mode} mode "
                         f"(split_len {split_len}): the pair-once mode shards over 1, 2, 4 or 8 ranks")
    return padded.value, chunk.value, split.value


def ring_schedule(rank: int, world_size: int):
    """[(hop, chunk sent to rank+1, chunk received from rank-1)] for hops 1..P-1 -- ``nbody_multi_ring_schedule``."""
    lib, out = _lib.load(), []
    for h in range(1, world_size):
        s, r = ctypes.c_int(0), ctypes.c_int(0)
        if lib.nbody_multi_ring_schedule(rank, world_size, h, ctypes.byref(s), ctypes.byref(r)) != _lib.NBODY_OK:
            raise ValueError("bad ring arguments")
        out.append((h, s.value, r.value))
    return out


def unique_id() -> bytes:
    """``nbody_multi_unique_id``: the RCCL id rank 0 creates and every rank of a multi-process system is given."""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _check(_lib.load().nbody_multi_unique_id(buf))
    return buf.raw


class MultiGpuSystem:
    """One process's view of a body set sharded over ``world_size`` GPUs; the interface of :class:`NBodySystem`."""

    def __init__(self, num_bodies: int, devices: Optional[Sequence[int]] = None, force_mode: str = "one_sided",
                 integrator: str = "kick_drift", exchange: str = "allgather", transport: str = "rccl", split_len: int = 0,
                 body_order: str = "given", create_timeout: float = 0.0,
                 _rank: Optional[int] = None, _world_size: Optional[int] = None, _unique_id: Optional[bytes] = None):
        if body_order not in _system.BODY_ORDERS:
            raise ValueError(f"body_order must be one of {_system.BODY_ORDERS}")
        self.body_order = body_order   # "morton": the library stores the state in nbody_morton_order, download() undoes it
        # create_timeout (seconds, rounded up; 0 = the library's default): the communicators are non-blocking and their creation
        # is polled under it -- a rank that never arrives makes the constructor raise instead of hanging in RCCL's bootstrap
        self._m = ctypes.c_void_p(None)
        self._lib = _lib.load()
        self.num_bodies = int(num_bodies)
        self.force_mode = "pair_once" if force_mode == "symmetric" else force_mode
        self.integrator, self.exchange, self.transport = integrator, exchange, transport
        cfg = _lib.MultiConfig(self.num_bodies, int(split_len), FORCE_MODES[force_mode], INTEGRATORS[integrator],
                               EXCHANGES[exchange], TRANSPORTS[transport], _system.BODY_ORDERS.index(body_order),
                               int(-(-float(create_timeout) // 1)) if create_timeout and create_timeout > 0 else 0)
        m = ctypes.c_void_p(None)
        if _rank is None:
            devs = list(devices if devices is not None else [0])
            arr = (ctypes.c_int * len(devs))(*devs)
            _check(self._lib.nbody_multi_create(ctypes.byref(m), ctypes.byref(cfg), arr, len(devs)))
            self.rank, self._devices = 0, devs
        else:
            dev = int((devices or [0])[0])
            ident = ctypes.create_string_buffer(_unique_id, UNIQUE_ID_BYTES)
            _check(self._lib.nbody_multi_create_rank(ctypes.byref(m), ctypes.byref(cfg), dev, int(_rank), int(_world_size), ident))
            self.rank, self._devices = int(_rank), [dev]
        self._m = m
        info = self.info()
        self.n_padded, self.chunk, self.split_len = info["n_padded"], info["rows_per_rank"], info["split_len"]
        self.world_size, self.local_ranks = info["world_size"], info["local_ranks"]
        self.kernels = self.shard(0)
        if force_mode == "auto":   # the library chose (NBODY_FORCE_AUTO): ask the shard context
            self.force_mode = {0: "one_sided", 1: "pair_once"}[int(self._lib.nbody_force_mode(self.kernels._ctx))]

    @classmethod
    def from_torch_distributed(cls, num_bodies: int, device: int, group=None, **kw) -> "MultiGpuSystem":
        """One rank per process: rank and world size come from ``torch.distributed`` (any backend -- it only carries the
        RCCL id from rank 0 to the others); every per-step exchange then runs inside the library."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(num_bodies, devices=[device], **kw)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(num_bodies, devices=[device], _rank=rank, _world_size=world, _unique_id=box[0], **kw)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_m", None) is not None and self._m.value:
            self._lib.nbody_multi_destroy(self._m)
            self._m = ctypes.c_void_p(None)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self) -> dict:
        out = (ctypes.c_int64 * 8)()
        _check(self._lib.nbody_multi_info(self._m, out), self._m)
        keys = ("n_bodies", "n_padded", "rows_per_rank", "split_len", "world_size", "local_ranks", "rccl_ranks", "exchange")
        return dict(zip(keys, (int(v) for v in out)))

    def shard(self, local_index: int = 0) -> _system.NBodySystem:
        """Local rank ``local_index``'s shard context as a borrowed :class:`NBodySystem` view: timing, device info,
        kernel selection.  Its buffers and its step belong to this object."""
        ctx = self._lib.nbody_multi_shard(self._m, int(local_index))
        if not ctx:
            raise IndexError("no such local rank")
        view = _system.NBodySystem.__new__(_system.NBodySystem)
        view._lib, view._ctx = self._lib, ctypes.c_void_p(ctx)
        view.num_bodies, view.split_len = self.n_padded, self.split_len
        view.close = lambda: None  # borrowed
        return view

    def positions_tensor(self, local_index: int = 0):
        """Local rank ``local_index``'s position replica as a zero-copy ``(n_padded, 4)`` float32 torch tensor on its
        device -- the mapped-pointer analogue of kernel.cu:1226 (a renderer reads it between steps).  Borrowed: valid
        until :meth:`close`; read it after :meth:`sync` (a step leaves the exchange of the updated rows in flight)."""
        return self._device_tensor(self._lib.nbody_multi_positions_device(self._m, int(local_index)), self.n_padded,
                                   int(local_index))

    def velocities_tensor(self, local_index: int = 0):
        """Local rank ``local_index``'s own velocity rows, ``(rows_per_rank, 4)``, zero-copy."""
        return self._device_tensor(self._lib.nbody_multi_velocities_device(self._m, int(local_index)), self.chunk,
                                   int(local_index))

    def _device_tensor(self, ptr, rows: int, local_index: int):
        import torch
        if not ptr or not rows:
            raise NBodyError(_lib.NBODY_ERR_STATE, "no such buffer")

        class _Borrowed:  # the CUDA array interface is how torch adopts foreign device memory without a copy
            __cuda_array_interface__ = {"shape": (int(rows), 4), "typestr": "<f4", "data": (int(ptr), False), "version": 3,
                                        "strides": None}
        return torch.as_tensor(_Borrowed(), device=torch.device("cuda", self._devices[local_index]))

    def set_timeout(self, seconds: float) -> None:
        _check(self._lib.nbody_multi_set_timeout(self._m, float(seconds)), self._m)

    def timing(self, on: bool = True) -> None:
        """Bracket the exchanges with HIP events and let the shard contexts time their kernels (``nbody_multi_timing_*``)."""
        _check(self._lib.nbody_multi_timing_enable(self._m, 1 if on else 0), self._m)

    def read_timing(self, local_index: int = 0) -> dict:
        """Totals of local rank ``local_index`` since the last read (sums of event-pair durations in ms, and counts): the
        kernels of its shard context, the position exchange on the communication stream and as the waiting force launch
        saw it, the pair-once column-sum exchange, layout refreshes, and the host time spent enqueuing the steps."""
        out = (ctypes.c_double * 16)()
        _check(self._lib.nbody_multi_timing_read(self._m, int(local_index), out), self._m)
        keys = ("steps", "host_enqueue_ms", "force_ms", "force_launches", "update_ms", "update_launches", "aux_ms", "aux_launches",
                "pos_exchange_comm_ms", "pos_exchanges", "pos_exchange_wait_ms", "pos_exchange_waits",
                "column_sum_exchange_ms", "column_sum_exchanges", "reorder_ms", "reorders")
        ints = {"steps", "force_launches", "update_launches", "aux_launches", "pos_exchanges", "pos_exchange_waits",
                "column_sum_exchanges", "reorders"}
        return {k: (int(v) if k in ints else float(v)) for k, v in zip(keys, out)}

    # -- buffers (kernel.cu:163-188) -----------------------------------------------------------------
    def reorder(self) -> None:
        """``body_order="morton"``: a new curve through the current positions (the layout decays as the bodies move); on the
        device, every rank from its own replica."""
        _check(self._lib.nbody_multi_reorder(self._m), self._m)

    def set_reorder_period(self, steps: int) -> None:
        """Refresh the layout by itself every ``steps`` steps (0: never)."""
        _check(self._lib.nbody_multi_set_reorder_period(self._m, int(steps)), self._m)

    @property
    def order(self):
        """``order[k]`` = the caller's index of the body in slot ``k`` of the replicas (``nbody_multi_order``); ``None`` when
        the bodies are stored as given."""
        if self.body_order == "given":
            return None
        perm = np.empty(self.num_bodies, dtype=np.int64)
        _check(self._lib.nbody_multi_order(self._m, perm.ctypes.data_as(ctypes.c_void_p)), self._m)
        return perm

    def _rows(self, data) -> np.ndarray:
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] != self.num_bodies:
            raise ValueError(f"expected {self.num_bodies} bodies, got {a.shape[0]}")
        return a

    def setParticlesPosition(self, data) -> None:
        """Host ``float4 {x,y,z,mass}`` of ALL bodies (the same on every process).  An independent copy, as in the reference:
        the velocities on the device stay with their bodies (``nbody_multi_set_positions``)."""
        a = self._rows(data)
        _check(self._lib.nbody_multi_set_positions(self._m, a.ctypes.data_as(ctypes.c_void_p)), self._m)

    def setParticlesVelocity(self, data) -> None:
        """Host ``float4 {vx,vy,vz,eps}`` of ALL bodies (each rank keeps its own rows); the positions on the device are not
        touched (``nbody_multi_set_velocities``)."""
        a = self._rows(data)
        _check(self._lib.nbody_multi_set_velocities(self._m, a.ctypes.data_as(ctypes.c_void_p)), self._m)

    def set_state(self, positions, velocities) -> None:
        p, v = self._rows(positions), self._rows(velocities)
        _check(self._lib.nbody_multi_set_state(self._m, p.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)),
               self._m)

    set_particles_position = setParticlesPosition
    set_particles_velocity = setParticlesVelocity

    def set_particle_softening(self, eps) -> None:
        if eps is None:
            _check(self._lib.nbody_multi_set_particle_softening(self._m, None), self._m)
            return
        e = np.ascontiguousarray(eps, dtype=np.float32).reshape(-1)
        if e.shape[0] != self.num_bodies:
            raise ValueError(f"expected {self.num_bodies} softening lengths, got {e.shape[0]}")
        _check(self._lib.nbody_multi_set_particle_softening(self._m, e.ctypes.data_as(ctypes.c_void_p)), self._m)

    def download(self) -> Tuple[np.ndarray, np.ndarray]:
        """(positions, velocities) of the real bodies, complete on every process."""
        p = np.empty((self.num_bodies, 4), dtype=np.float32)
        v = np.empty((self.num_bodies, 4), dtype=np.float32)
        _check(self._lib.nbody_multi_download(self._m, p.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)),
               self._m)
        return p, v

    # -- the step (kernel.cu:1225-1242) ---------------------------------------------------------------
    def step(self, dt: float = _system.TIME_TICK, softening: float = _system.SOFTENING_VERSION3, sync: bool = True) -> None:
        fn = self._lib.nbody_multi_step if sync else self._lib.nbody_multi_step_async
        _check(fn(self._m, float(dt), float(softening)), self._m)

    def step_n(self, n: int, dt: float = _system.TIME_TICK, softening: float = _system.SOFTENING_VERSION3) -> None:
        _check(self._lib.nbody_multi_step_n(self._m, int(n), float(dt), float(softening)), self._m)

    def sync(self) -> None:
        _check(self._lib.nbody_multi_sync(self._m), self._m)

    # -- diagnostics -----------------------------------------------------------------------------------
    def energy(self, softening: float) -> np.ndarray:
        out = (ctypes.c_double * 3)()
        _check(self._lib.nbody_multi_energy(self._m, float(softening), out), self._m)
        return np.array(list(out), dtype=np.float64)

    def momentum(self) -> np.ndarray:
        out = (ctypes.c_double * 4)()
        _check(self._lib.nbody_multi_momentum(self._m, out), self._m)
        return np.array(list(out), dtype=np.float64)

    def replicas_identical(self) -> bool:
        """Every rank (of every process) holds the same position bits."""
        out = (ctypes.c_uint64 * 2)()
        _check(self._lib.nbody_multi_replica_checksums(self._m, out), self._m)
        return int(out[0]) == int(out[1])


def sharded_system(num_bodies: int, device: int = 0, group=None, exchange: str = "allgather", force_mode: str = "one_sided",
                   integrator: str = "kick_drift", split_len: int = 0, body_order: str = "given") -> MultiGpuSystem:
    """The sharded system of this process's rank: one rank per process when a ``torch.distributed`` process group exists
    (whatever its backend: it only carries the RCCL id), a single-rank system otherwise.  The exchange is the library's."""
    return MultiGpuSystem.from_torch_distributed(num_bodies, device, group=group, exchange=exchange, force_mode=force_mode,
                                                 integrator=integrator, split_len=split_len, body_order=body_order)
