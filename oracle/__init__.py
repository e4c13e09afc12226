"""ctypes front end of the CPU oracle (``nbody_oracle.c``).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED -- see the header of ``nbody_oracle.c``: the reference has no tests or golden
vectors for this path and cannot be built here, so this restatement is pinned by analytic known
answers only.

Importers allowed: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``n_body_problem_amd/`` imports this package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnbody_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_c_int = ctypes.c_int


def build(force: bool = False) -> str:
    """Compile ``libnbody_oracle.so`` with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libnbody_oracle.so"], check=True, capture_output=True)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_accel_f32.argtypes = [_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _f32p, _c_int]
        L.oracle_accel_f32.restype = None
        L.oracle_accel_f64.argtypes = [_f64p, _c_int, _c_int, _c_int, _c_int, ctypes.c_double, _f64p, _c_int]
        L.oracle_accel_f64.restype = None
        L.oracle_accel_f64_from_f32.argtypes = [_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                                                _f64p, _c_int]
        L.oracle_accel_f64_from_f32.restype = _c_int
        L.oracle_accel_f64_pps.argtypes = [_f64p, _f64p, _c_int, _c_int, _c_int, ctypes.c_double, _f64p]
        L.oracle_accel_f64_pps.restype = None
        L.oracle_potential_pps.argtypes = [_f64p, _f64p, _c_int, ctypes.c_double]
        L.oracle_potential_pps.restype = ctypes.c_double
        L.oracle_update_f32.argtypes = [_f32p, _f32p, _f32p, _c_int, _c_int, ctypes.c_float]
        L.oracle_update_f32.restype = None
        L.oracle_step_f32.argtypes = [_f32p, _f32p, _c_int, ctypes.c_float, ctypes.c_float, _c_int, _c_int]
        L.oracle_step_f32.restype = _c_int
        L.oracle_step_kdk_f32.argtypes = [_f32p, _f32p, _c_int, ctypes.c_float, ctypes.c_float, _c_int, _c_int]
        L.oracle_step_kdk_f32.restype = _c_int
        L.oracle_step_f64.argtypes = [_f64p, _f64p, _c_int, ctypes.c_double, ctypes.c_double, _c_int, _c_int]
        L.oracle_step_f64.restype = _c_int
        L.oracle_step_v3.argtypes = [_f32p, _f32p, _c_int, _c_int]
        L.oracle_step_v3.restype = _c_int
        L.oracle_step_v2_serial.argtypes = [_f32p, _f32p, _c_int, _c_int]
        L.oracle_step_v2_serial.restype = None
        L.oracle_pair_v3.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_pair_v3.restype = None
        L.oracle_pair_v1.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_pair_v1.restype = None
        L.oracle_energy.argtypes = [_f32p, _f32p, _c_int, ctypes.c_float, _f64p, _c_int]
        L.oracle_energy.restype = _c_int
        L.oracle_momentum.argtypes = [_f32p, _f32p, _c_int, _f64p]
        L.oracle_momentum.restype = None
        _lib = L
    return _lib


def cpus_visible() -> int:
    """CPUs this process may be scheduled on (``sched_getaffinity``)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


def cpu_quota_cores():
    """CPU time per second of wall time the cgroup grants (v2 ``cpu.max`` / v1 cfs quota), in cores; None = unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except Exception:
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except Exception:
        return None


def host_threads() -> int:
    """Worker threads worth starting: the visible CPUs, capped by the cgroup's CPU quota (a GPU box shows 256 CPUs and
    grants 16 cores' worth of time; 256 threads on it run slower than 16)."""
    quota = cpu_quota_cores()
    visible = cpus_visible()
    return max(1, min(visible, int(round(quota)))) if quota else visible


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def accel_f32(pos, i0=0, i1=None, j0=0, j1=None, eps=1e-3, threads=None) -> np.ndarray:
    """Reference-order fp32 accelerations of rows [i0,i1) from columns [j0,j1): (i1-i0, 3) float32."""
    pos = _f32(pos).reshape(-1, 4)
    n = pos.shape[0]
    i1 = n if i1 is None else i1
    j1 = n if j1 is None else j1
    out = np.zeros((max(i1 - i0, 0), 3), dtype=np.float32)
    if i1 > i0:
        lib().oracle_accel_f32(pos, i0, i1, j0, j1, float(eps), out, threads or host_threads())
    return out


def accel_f64(pos, i0=0, i1=None, j0=0, j1=None, eps=1e-3, threads=None) -> np.ndarray:
    """fp64-truth accelerations from fp32 (or fp64) positions: (i1-i0, 3) float64."""
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 4)
    n = p.shape[0]
    i1 = n if i1 is None else i1
    j1 = n if j1 is None else j1
    out = np.zeros((max(i1 - i0, 0), 3), dtype=np.float64)
    if i1 > i0:
        lib().oracle_accel_f64(p, i0, i1, j0, j1, float(eps), out, threads or host_threads())
    return out


def accel_f64_pps(pos, eps_pp, eps=0.0, i0=0, i1=None) -> np.ndarray:
    """fp64 accelerations with per-particle softening: eps_ij^2 = eps^2 + eps_i^2 + eps_j^2."""
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 4)
    e = np.ascontiguousarray(eps_pp, dtype=np.float64).reshape(-1)
    n = p.shape[0]
    i1 = n if i1 is None else i1
    out = np.zeros((i1 - i0, 3), dtype=np.float64)
    lib().oracle_accel_f64_pps(p, e, n, i0, i1, float(eps), out)
    return out


def potential_pps(pos, eps_pp, eps=0.0) -> float:
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 4)
    e = np.ascontiguousarray(eps_pp, dtype=np.float64).reshape(-1)
    return float(lib().oracle_potential_pps(p, e, p.shape[0], float(eps)))


def step_f32(pos, vel, dt, eps, nsteps=1, threads=None):
    """nsteps reference-order fp32 steps; returns new (pos, vel) copies, (n,4) float32 each."""
    p = _f32(pos).reshape(-1, 4).copy()
    v = _f32(vel).reshape(-1, 4).copy()
    rc = lib().oracle_step_f32(p, v, p.shape[0], float(dt), float(eps), int(nsteps), threads or host_threads())
    if rc != 0:
        raise MemoryError("oracle_step_f32")
    return p, v


def step_kdk_f32(pos, vel, dt, eps, nsteps=1, threads=None):
    """nsteps kick-drift-kick (velocity Verlet) steps in reference-order fp32; returns new (pos, vel)."""
    p = _f32(pos).reshape(-1, 4).copy()
    v = _f32(vel).reshape(-1, 4).copy()
    rc = lib().oracle_step_kdk_f32(p, v, p.shape[0], float(dt), float(eps), int(nsteps), threads or host_threads())
    if rc != 0:
        raise MemoryError("oracle_step_kdk_f32")
    return p, v


def step_f64(pos, vel, dt, eps, nsteps=1, threads=None):
    """nsteps fp64-truth steps with the state kept in double; returns (pos, vel) as (n,4) float64.

    ``dt`` and ``eps`` are first rounded to fp32 (the values step() actually receives)."""
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 4).copy()
    v = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 4).copy()
    rc = lib().oracle_step_f64(p, v, p.shape[0], float(np.float32(dt)), float(np.float32(eps)), int(nsteps),
                               threads or host_threads())
    if rc != 0:
        raise MemoryError("oracle_step_f64")
    return p, v


def update_f32(pos, vel, acc3, dt, i0=0, i1=None):
    """In-place kick-drift of rows [i0,i1) (kernel.cu:777-801); acc3 is (i1-i0, 3)."""
    n = pos.shape[0]
    i1 = n if i1 is None else i1
    lib().oracle_update_f32(pos, vel, _f32(acc3), i0, i1, float(dt))


def step_v3(pos, vel, nsteps=1):
    """The reference's VERSION 3 with its own constants (dt=0.008, effective eps=1e-2), deterministic order."""
    p = _f32(pos).reshape(-1, 4).copy()
    v = _f32(vel).reshape(-1, 4).copy()
    if lib().oracle_step_v3(p, v, p.shape[0], int(nsteps)) != 0:
        raise MemoryError("oracle_step_v3")
    return p, v


def step_v2_serial(pos, vel, nsteps=1):
    """The reference's VERSION 2 (Gauss-Seidel, dt=0.008, eps=1e-3)."""
    p = _f32(pos).reshape(-1, 4).copy()
    v = _f32(vel).reshape(-1, 4).copy()
    lib().oracle_step_v2_serial(p, v, p.shape[0], int(nsteps))
    return p, v


def pair_v3(a, b) -> np.ndarray:
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_pair_v3(_f32(a), _f32(b), out)
    return out


def pair_v1(a, b, acc=None) -> np.ndarray:
    out = np.zeros(3, dtype=np.float32) if acc is None else _f32(acc).copy()
    lib().oracle_pair_v1(_f32(a), _f32(b), out)
    return out


def energy(pos, vel, eps, threads=None) -> np.ndarray:
    """[kinetic, potential, total] in fp64 (Plummer-softened potential, G=1)."""
    p = _f32(pos).reshape(-1, 4)
    v = _f32(vel).reshape(-1, 4)
    out = np.zeros(3, dtype=np.float64)
    if lib().oracle_energy(p, v, p.shape[0], float(eps), out, threads or host_threads()) != 0:
        raise MemoryError("oracle_energy")
    return out


def momentum(pos, vel) -> np.ndarray:
    """[px, py, pz, total mass] in fp64."""
    p = _f32(pos).reshape(-1, 4)
    v = _f32(vel).reshape(-1, 4)
    out = np.zeros(4, dtype=np.float64)
    lib().oracle_momentum(p, v, p.shape[0], out)
    return out
