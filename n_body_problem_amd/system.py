"""Host-side mirror of the reference's step interface, over the C ABI of ``libnbody_amd.so``.

The reference has no library API; its "operator interface" for this path is (main_project/kernel.cu):

* ``initialize(numBodies)``                       :130-161  allocate the position VBO + velocity buffer
* ``setParticlesPosition(data)``                  :163-177  host float4 {x,y,z,mass} -> device
* ``setParticlesVelocity(data)``                  :179-188  host float4 {vx,vy,vz,eps} -> device
* the per-frame bracket                           :1225-1242 = ``step(positions, velocities, masses, dt, softening)``

:class:`NBodySystem` keeps those names (plus snake_case aliases) and the same argument meaning.
Device memory and streams come from PyTorch-ROCm (plumbing only); all arithmetic happens in the
hand-written HIP kernels behind ``include/nbody.h``.  There is no CPU or PyTorch fallback: without
the built library or without a gfx950 device every constructor raises.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import NBodyError, check

#: the reference's compile-time constants (kernel.cu:63, :66 and SURVEY.md 8a for the effective values)
TIME_TICK = 0.008
SOFTENING_VERSION3 = 1.0e-2  # cal_single_acclerate_without_mass_new: 0.1 pre-scale => eps^2 = 1e-4
SYM_GROUPS = 8                   # NBODY_SYM_GROUPS
SOFTENING_VERSION1 = 1.0e-3  # cal_single_acclerate: eps^2 = EPSILON = 1e-6
BLOCK_SIZE = 256


def _torch():
    import torch  # imported lazily: the ctypes layer itself does not need torch
    return torch


def _ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def pair_once_split_len(n_total: int) -> int:
    """The split length to create a pair-once system with (a function of ``n_total`` only)."""
    return int(_lib.load().nbody_pair_once_split_len(int(n_total)))


def default_split_len(n_total: int) -> int:
    """Columns per partial sum for ``n_total`` bodies (a function of ``n_total`` only)."""
    return int(_lib.load().nbody_default_split_len(int(n_total)))


BODY_ORDERS = ("given", "morton")


def morton_order(positions) -> np.ndarray:
    """``perm[k]`` = index of the body to store at slot ``k`` (``nbody_morton_order``: along a Morton curve, the bodies of one
    mass together when there are few distinct masses).  Neighbours in the arrays become neighbours in space, the force
    kernels' operands toggle fewer bits and the power-limited clock rises: 3.7 % per force pass at N = 2^20."""
    a = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 4)
    perm = np.empty(a.shape[0], dtype=np.int64)
    status = _lib.load().nbody_morton_order(a.ctypes.data_as(ctypes.c_void_p), a.shape[0], perm.ctypes.data_as(ctypes.c_void_p))
    if status != 0:
        raise NBodyError(status, "nbody_morton_order failed")
    return perm


class NBodySystem:
    """One context = one GPU's rows ``[row_lo, row_lo+row_count)`` against all ``num_bodies`` columns.

    With the defaults it is the reference's single-GPU system: ``initialize(numBodies)``.

    ``body_order="morton"`` (a context that owns every row): the device tensors hold the bodies in :func:`morton_order` of
    their positions -- laid and refreshed ON THE DEVICE (``nbody_reorder``, csrc/nbody_order.hip: the same permutation as
    the host function, without a host copy of the state) -- the setters and :meth:`download` speak the caller's order, and
    ``self.order[k]`` is the caller's index of the body in slot ``k`` of the device tensors.  The physics is the same, the
    sums are taken in another order.
    """

    def __init__(self, num_bodies: int, device: int = 0, row_lo: int = 0, row_count: Optional[int] = None,
                 split_len: int = 0, body_order: str = "given"):
        if body_order not in BODY_ORDERS:
            raise ValueError(f"body_order must be one of {BODY_ORDERS}")
        self.body_order = body_order
        self._order_dev = None   # body_order="morton": int64 device tensor, slot k holds the caller's body _order_dev[k]
        self._order_host = None  # its host copy, fetched when asked for
        self._reorder_period, self._steps_since_order = 0, 0
        self._ctx = ctypes.c_void_p(None)
        self._lib = _lib.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise NBodyError(_lib.NBODY_ERR_NO_DEVICE, "no HIP device visible to PyTorch; there is no CPU path")
        self.num_bodies = int(num_bodies)
        self.row_lo = int(row_lo)
        self.row_count = self.num_bodies - self.row_lo if row_count is None else int(row_count)
        self.device = torch.device("cuda", device)
        ctx = ctypes.c_void_p(None)
        check(self._lib.nbody_create_shard(ctypes.byref(ctx), device, self.num_bodies, self.row_lo, self.row_count,
                                           int(split_len)), None)
        self._ctx = ctx
        self.split_len = int(self._lib.nbody_split_len(ctx))
        if body_order != "given" and (self.row_lo != 0 or self.row_count != self.num_bodies):
            self.close()
            raise ValueError("body_order needs a context that owns every row (shards: MultiGpuSystem(body_order=...))")
        self._eps_pp = None
        # the reference's two device buffers: position "VBO" (all bodies) and velocities (own rows)
        self.positions = torch.zeros((self.num_bodies, 4), dtype=torch.float32, device=self.device)
        self.velocities = torch.zeros((self.row_count, 4), dtype=torch.float32, device=self.device)
        if body_order == "morton":
            self._order_dev = torch.empty((self.num_bodies,), dtype=torch.int64, device=self.device)
            self._use_current_stream()
            check(self._lib.nbody_order_identity(self._ctx, _ptr(self._order_dev), self.num_bodies), self._ctx)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.nbody_destroy(self._ctx)
            self._ctx = ctypes.c_void_p(None)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _use_current_stream(self) -> None:
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.nbody_set_stream(self._ctx, ctypes.c_void_p(s)), self._ctx)

    # -- body order ------------------------------------------------------------------------------
    @property
    def order(self):
        """``order[k]`` = the caller's index of the body in slot ``k`` of the device tensors (``None``: stored as given)."""
        if self._order_dev is None:
            return None
        if self._order_host is None:
            self._order_host = self._order_dev.cpu().numpy()
        return self._order_host

    def _to_slots(self, dst, rows, floats_per_row: int) -> None:
        """``dst[k] = rows[order[k]]`` on the device: ``rows`` (a device tensor in the caller's order) into the slots."""
        self._use_current_stream()
        check(self._lib.nbody_order_set(self._ctx, _ptr(self._order_dev), self.num_bodies, 0), self._ctx)
        check(self._lib.nbody_order_gather(self._ctx, _ptr(dst), _ptr(rows), 0, self.num_bodies, floats_per_row), self._ctx)

    def _lay_curve(self) -> None:
        """``nbody_reorder``: positions, velocities, softening lengths and the order array into :func:`morton_order` of the
        positions the device holds now."""
        self._use_current_stream()
        check(self._lib.nbody_reorder(self._ctx, _ptr(self.positions), _ptr(self.velocities), _ptr(self._eps_pp),
                                      _ptr(self._order_dev), self.num_bodies), self._ctx)
        self._order_host = None
        self._steps_since_order = 0

    # -- buffers (kernel.cu:163-188) ----------------------------------------------------------
    def setParticlesPosition(self, data) -> None:
        """Host ``float4 {x,y,z,mass}`` for ALL bodies -> the device position buffer.  An independent copy, as in the
        reference: velocities and softening lengths already set stay with their bodies."""
        torch = _torch()
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] != self.num_bodies:
            raise ValueError(f"expected {self.num_bodies} bodies, got {a.shape[0]}")
        if self.body_order == "morton":
            rows = torch.from_numpy(a).to(self.device)   # the caller's order; each body's position to the body's slot
            self._to_slots(self.positions, rows, 4)
            self._lay_curve()                            # then the curve through the new positions: everything follows
            torch.cuda.current_stream(self.device).synchronize()   # `rows` is released behind the gather
        else:
            self.positions.copy_(torch.from_numpy(a))
        self._lib.nbody_invalidate_forces(self._ctx)

    def setParticlesVelocity(self, data) -> None:
        """Host ``float4 {vx,vy,vz,eps}`` -> the device velocity buffer.

        Accepts all ``num_bodies`` rows (this context's slice is taken) or exactly its own rows."""
        torch = _torch()
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] == self.num_bodies:
            a = a[self.row_lo:self.row_lo + self.row_count]
        if a.shape[0] != self.row_count:
            raise ValueError(f"expected {self.row_count} or {self.num_bodies} velocity rows, got {a.shape[0]}")
        if self.body_order == "morton":
            rows = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._to_slots(self.velocities, rows, 4)
            torch.cuda.current_stream(self.device).synchronize()
        else:
            self.velocities.copy_(torch.from_numpy(np.ascontiguousarray(a)))

    set_particles_position = setParticlesPosition
    set_particles_velocity = setParticlesVelocity

    def set_reorder_period(self, steps: int) -> None:
        """``body_order="morton"``: :meth:`step_n` refreshes the layout by itself every ``steps`` steps (0: never) -- the
        same schedule as ``nbody_multi_set_reorder_period``, so a single context and shards keep producing the same bits."""
        if steps < 0:
            raise ValueError("steps must be >= 0")
        self._reorder_period = int(steps)

    def reorder(self) -> None:
        """``body_order="morton"``: a new curve through the current positions (the layout decays as the bodies move: at
        N = 2^20 half of the gain is gone after ~300 steps of dt = 1e-3).  On the device (``nbody_reorder``), asynchronous."""
        self._steps_since_order = 0
        if self.body_order == "morton":
            self._lay_curve()

    def download(self) -> Tuple[np.ndarray, np.ndarray]:
        """(positions, velocities) as host float32 arrays, in the caller's body order."""
        pos, vel = self.positions.cpu().numpy(), self.velocities.cpu().numpy()
        order = self.order
        if order is not None:
            p, v = np.empty_like(pos), np.empty_like(vel)
            p[order], v[order] = pos, vel
            return p, v
        return pos, vel

    # -- the step (kernel.cu:1225-1242) ---------------------------------------------------------
    def step(self, dt: float = TIME_TICK, softening: float = SOFTENING_VERSION3, masses=None, sync: bool = True):
        """One step in place on ``self.positions`` / ``self.velocities``.

        ``masses`` (optional device tensor of ``num_bodies`` floats) is first copied into
        ``positions[:,3]``; the default, as in the reference, is that mass already lives there."""
        self._use_current_stream()
        if self._reorder_period > 0 and self.body_order == "morton" and self._steps_since_order >= self._reorder_period:
            self.reorder()
        self._steps_since_order += 1
        fn = self._lib.nbody_step if sync else self._lib.nbody_step_async
        if masses is not None and self._order_dev is not None:   # the caller's order -> the slots
            slots = _torch().empty_like(masses.reshape(-1))
            self._to_slots(slots, masses.reshape(-1).contiguous(), 1)
            masses = slots
        check(fn(self._ctx, _ptr(self.positions), _ptr(self.velocities), _ptr(masses), float(dt), float(softening)),
              self._ctx)

    def step_n(self, k: int, dt: float = TIME_TICK, softening: float = SOFTENING_VERSION3) -> None:
        """``k`` steps enqueued back to back, one synchronisation at the end (``nbody_step_n_on``: small systems replay a
        captured HIP graph of one step, see :meth:`set_graph_replay`)."""
        self._use_current_stream()
        k = int(k)
        if k <= 0:
            check(self._lib.nbody_step_n_on(self._ctx, _ptr(self.positions), _ptr(self.velocities), k, float(dt),
                                            float(softening)), self._ctx)
        while k > 0:
            if self._reorder_period > 0 and self.body_order == "morton" and self._steps_since_order >= self._reorder_period:
                self.reorder()
            run = k if self._reorder_period <= 0 or self.body_order != "morton" else min(k, self._reorder_period - self._steps_since_order)
            check(self._lib.nbody_step_n_on(self._ctx, _ptr(self.positions), _ptr(self.velocities), run, float(dt),
                                            float(softening)), self._ctx)
            self._steps_since_order += run
            k -= run

    def set_strip_len(self, length: int) -> None:
        """Pair-once mode: column splits a tile workgroup takes with the rows' sums kept in registers (``nbody_set_strip_len``;
        0 = automatic: 4 with 2048-body splits, i.e. from N = 2^20)."""
        check(self._lib.nbody_set_strip_len(self._ctx, int(length)), self._ctx)

    def set_graph_replay(self, mode: int) -> None:
        """-1: automatic (a pair-once step of more than two kernels up to 32 768 bodies, where the replay measured faster), 0: never, 1: always."""
        check(self._lib.nbody_set_graph_replay(self._ctx, int(mode)), self._ctx)

    def forces(self, col_lo: int, col_count: int, softening: float, positions=None) -> None:
        """Partial accelerations of this context's rows from columns ``[col_lo, col_lo+col_count)`` (async)."""
        self._use_current_stream()
        p = self.positions if positions is None else positions
        check(self._lib.nbody_forces(self._ctx, _ptr(p), int(col_lo), int(col_count), float(softening)), self._ctx)

    def forces_complement(self, col_lo: int, col_count: int, softening: float, positions=None) -> None:
        """Partial accelerations from every column EXCEPT ``[col_lo, col_lo+col_count)``, one launch (async)."""
        self._use_current_stream()
        p = self.positions if positions is None else positions
        check(self._lib.nbody_forces_complement(self._ctx, _ptr(p), int(col_lo), int(col_count), float(softening)),
              self._ctx)

    def update(self, dt: float, positions=None, velocities=None) -> None:
        """Sum the partials of all splits and kick-drift this context's rows (async)."""
        self._use_current_stream()
        p = self.positions if positions is None else positions
        v = self.velocities if velocities is None else velocities
        check(self._lib.nbody_update(self._ctx, _ptr(p), _ptr(v), float(dt)), self._ctx)

    def sync(self) -> None:
        check(self._lib.nbody_sync(self._ctx), self._ctx)

    # -- integrator (kick-drift = the reference's final scheme; kdk = velocity Verlet with cached accelerations) ----
    def set_integrator(self, name: str) -> None:
        check(self._lib.nbody_set_integrator(self._ctx, {"kick_drift": 0, "kdk": 1}[name]), self._ctx)

    def invalidate_forces(self) -> None:
        check(self._lib.nbody_invalidate_forces(self._ctx), self._ctx)

    def kdk_prepare(self) -> None:
        self._use_current_stream()
        check(self._lib.nbody_kdk_prepare(self._ctx), self._ctx)

    def kdk_kick_drift(self, dt: float, positions=None, velocities=None) -> None:
        self._use_current_stream()
        p = self.positions if positions is None else positions
        v = self.velocities if velocities is None else velocities
        check(self._lib.nbody_kdk_kick_drift(self._ctx, _ptr(p), _ptr(v), float(dt)), self._ctx)

    def kdk_kick(self, dt: float, velocities=None) -> None:
        self._use_current_stream()
        v = self.velocities if velocities is None else velocities
        check(self._lib.nbody_kdk_kick(self._ctx, _ptr(v), float(dt)), self._ctx)

    # -- diagnostics -----------------------------------------------------------------------------
    def energy(self, softening: float) -> np.ndarray:
        """[kinetic, potential, total] of this context's rows (fp64)."""
        self._use_current_stream()
        out = (ctypes.c_double * 3)()
        check(self._lib.nbody_energy(self._ctx, _ptr(self.positions), _ptr(self.velocities), float(softening), out),
              self._ctx)
        return np.array(list(out), dtype=np.float64)

    def momentum(self) -> np.ndarray:
        """[px, py, pz, mass] of this context's rows (fp64)."""
        self._use_current_stream()
        out = (ctypes.c_double * 4)()
        check(self._lib.nbody_momentum(self._ctx, _ptr(self.positions), _ptr(self.velocities), out), self._ctx)
        return np.array(list(out), dtype=np.float64)

    # -- measurement -----------------------------------------------------------------------------
    def timing(self, on: bool = True) -> None:
        check(self._lib.nbody_timing_enable(self._ctx, 1 if on else 0), self._ctx)

    def read_timing(self) -> dict:
        """HIP-event totals since the last read: sums of per-launch durations (ms) and launch counts of the force kernel,
        the update kernels and, in the pair-once mode, the diagonal-tile kernel (``aux``)."""
        out = (ctypes.c_double * 6)()
        check(self._lib.nbody_timing_read_ex(self._ctx, out), self._ctx)
        return {"force_ms": out[0], "force_launches": int(out[1]), "update_ms": out[2], "update_launches": int(out[3]),
                "aux_ms": out[4], "aux_launches": int(out[5])}

    def set_force_mode(self, mode: str) -> None:
        """``"one_sided"`` (default) or ``"symmetric"`` (the pair-once kernel; create the system with
        ``split_len=pair_once_split_len(n)``).  A shard in the pair-once mode exchanges ``self.colparts`` once per step:
        ``forces*`` -> ``sym_reduce()`` -> all-gather of ``sym_own_slice()`` -> ``update``."""
        code = {"one_sided": 0, "symmetric": 1, "pair_once": 1, "auto": 2}[mode]
        check(self._lib.nbody_set_force_mode(self._ctx, code), self._ctx)
        self.split_len = int(self._lib.nbody_split_len(self._ctx))   # "auto" picks the split length of the mode it picks
        self.force_mode = {0: "one_sided", 1: "pair_once"}[int(self._lib.nbody_force_mode(self._ctx))]
        code = 1 if self.force_mode == "pair_once" else 0
        self.colparts = None
        if code == 1 and (self.row_lo != 0 or self.row_count != self.num_bodies):
            torch = _torch()  # the exchange buffer lives in a tensor so that torch.distributed can gather into it
            self.colparts = torch.zeros((SYM_GROUPS, self.num_bodies, 4), dtype=torch.float32, device=self.device)
            check(self._lib.nbody_sym_set_colparts(self._ctx, _ptr(self.colparts)), self._ctx)

    def sym_groups(self):
        """(first group, group count, splits per group) of this context in the pair-once summation order."""
        lo, cnt, gs = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.nbody_sym_groups(self._ctx, ctypes.byref(lo), ctypes.byref(cnt), ctypes.byref(gs)), self._ctx)
        return lo.value, cnt.value, gs.value

    def sym_own_slice(self):
        """The slices of ``colparts`` this context writes (a view): what it contributes to the all-gather."""
        lo, cnt, _ = self.sym_groups()
        return self.colparts[lo:lo + cnt]

    def sym_reduce(self) -> None:
        self._use_current_stream()
        check(self._lib.nbody_sym_reduce(self._ctx), self._ctx)

    def set_particle_softening(self, eps) -> None:
        """Per-particle softening lengths for ALL bodies (host array or device tensor of ``num_bodies`` floats), e.g.
        the ``vel[:,3]`` the reference's loaders fill; ``None`` switches it off.  eps_ij^2 = softening^2 + eps_i^2 + eps_j^2."""
        torch = _torch()
        if eps is None:
            self._eps_pp = None
        else:
            t = eps if torch.is_tensor(eps) else torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32))
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if t.numel() != self.num_bodies:
                raise ValueError(f"expected {self.num_bodies} softening lengths, got {t.numel()}")
            t = t.reshape(-1)
            if self._order_dev is not None:  # the caller's order -> the order of the device buffers
                slots = torch.empty_like(t)
                self._to_slots(slots, t, 1)
                torch.cuda.current_stream(self.device).synchronize()
                t = slots
            self._eps_pp = t  # keeps the borrowed device buffer alive
        check(self._lib.nbody_set_particle_softening(self._ctx, _ptr(self._eps_pp)), self._ctx)

    def set_rows_per_lane(self, rpl: int) -> None:
        check(self._lib.nbody_set_rows_per_lane(self._ctx, int(rpl)), self._ctx)

    def set_equal_mass_path(self, on: bool) -> None:
        """Splits whose bodies all carry one mass leave the mass out of the inner loop (on by default); ``False`` sends
        every split down the general path (A/B measurement, tests)."""
        check(self._lib.nbody_set_equal_mass_path(self._ctx, 1 if on else 0), self._ctx)

    def set_early_summation(self, on: bool) -> None:
        """Pair-once mode on one context: sum the finished row groups beside the last group's tiles (on by default; the
        result is bit-identical either way)."""
        check(self._lib.nbody_set_early_summation(self._ctx, 1 if on else 0), self._ctx)

    def set_summation_parts(self, parts: int) -> None:
        """Pair-once mode on one context: 1, 2, 4 or 8 launches of whole row groups, each part's sums formed beside the next
        part's tiles; 4 and 8 keep only two parts' partial sums in memory.  0 = the default.  Bit-identical throughout."""
        check(self._lib.nbody_set_summation_parts(self._ctx, int(parts)), self._ctx)

    def partial_sum_bytes(self) -> int:
        """Bytes of partial-sum arrays allocated so far (the first force call allocates them)."""
        return int(self._lib.nbody_partial_sum_bytes(self._ctx))

    def device_info(self) -> dict:
        out = (ctypes.c_int64 * 4)()
        name = ctypes.create_string_buffer(128)
        check(self._lib.nbody_device_info(self._ctx, out, name, 128), self._ctx)
        return {"compute_units": out[0], "clock_mhz": out[1], "wavefront": out[2], "lds_per_cu": out[3],
                "name": name.value.decode()}


PAIR_ONCE_MIN_BODIES = 0  # NBODY_PAIR_ONCE_MIN_BODIES: since round 4 the pair-once step is the faster one at every size


def initialize(num_bodies: int, device: int = 0, force_mode: str = "one_sided", body_order: str = "given") -> NBodySystem:
    """``initialize(numBodies)`` of kernel.cu:130-161.  ``force_mode``: ``"one_sided"``, ``"pair_once"`` or ``"auto"``
    (pair-once from ``PAIR_ONCE_MIN_BODIES`` bodies on, where it is the faster one); ``body_order``: see :class:`NBodySystem`."""
    if force_mode == "auto":   # nbody_set_force_mode(ctx, NBODY_FORCE_AUTO): the choice is the library's, for every host language
        s = NBodySystem(num_bodies, device=device, body_order=body_order)
        s.set_force_mode("auto")
        return s
    s = NBodySystem(num_bodies, device=device, split_len=pair_once_split_len(num_bodies) if force_mode == "pair_once" else 0,
                    body_order=body_order)
    s.set_force_mode(force_mode)
    return s


_STEP_CACHE: dict = {}


def step(positions, velocities, masses=None, dt: float = TIME_TICK, softening: float = SOFTENING_VERSION3):
    """``step(positions, velocities, masses, dt, softening)`` on caller-owned device tensors, in place.

    ``positions``: (n,4) float32 CUDA tensor {x,y,z,mass}; ``velocities``: (n,4) float32 CUDA tensor;
    ``masses``: None (mass is ``positions[:,3]``, the reference's layout) or an (n,) float32 CUDA tensor.
    Synchronous, like the reference's bracket (kernel.cu:1232,1236)."""
    torch = _torch()
    if not (positions.is_cuda and velocities.is_cuda):
        raise NBodyError(_lib.NBODY_ERR_NO_DEVICE, "step() takes device tensors; there is no CPU path")
    if positions.dtype != torch.float32 or velocities.dtype != torch.float32:
        raise TypeError("positions and velocities must be float32")
    if not (positions.is_contiguous() and velocities.is_contiguous()):
        raise ValueError("positions and velocities must be contiguous (n,4) buffers")
    n = positions.shape[0]
    if positions.shape != (n, 4) or velocities.shape != (n, 4):
        raise ValueError("expected (n,4) positions and velocities")
    if masses is not None and (masses.dtype != torch.float32 or masses.numel() != n or not masses.is_cuda):
        raise ValueError("masses must be an (n,) float32 device tensor")
    key = (positions.device.index or 0, n)
    lib = _lib.load()
    ctx = _STEP_CACHE.get(key)
    if ctx is None:
        ctx = ctypes.c_void_p(None)
        check(lib.nbody_create(ctypes.byref(ctx), key[0], n), None)
        _STEP_CACHE[key] = ctx
    s = torch.cuda.current_stream(positions.device).cuda_stream
    check(lib.nbody_set_stream(ctx, ctypes.c_void_p(s)), ctx)
    check(lib.nbody_step(ctx, _ptr(positions), _ptr(velocities), _ptr(masses), float(dt), float(softening)), ctx)
