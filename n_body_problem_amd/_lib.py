"""ctypes binding of ``libnbody_amd.so`` -- one prototype per entry point of ``include/nbody.h``.

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

from .build import LIB_PATH

NBODY_OK = 0
NBODY_ERR_INVALID = -1
NBODY_ERR_ALLOC = -2
NBODY_ERR_DEVICE = -3
NBODY_ERR_NO_DEVICE = -4
NBODY_ERR_STATE = -5


class NBodyError(RuntimeError):
    """A C-ABI call returned a negative ``nbody_status``."""

    def __init__(self, status: int, message: str):
        super().__init__(f"nbody status {status}: {message}")
        self.status = status


_PROTOTYPES = {
    # name: (restype, argtypes)
    "nbody_abi_version": (c_int, []),
    "nbody_status_string": (c_char_p, [c_int]),
    "nbody_create": (c_int, [POINTER(c_void_p), c_int, c_int64]),
    "nbody_create_shard": (c_int, [POINTER(c_void_p), c_int, c_int64, c_int64, c_int64, c_int64]),
    "nbody_destroy": (c_int, [c_void_p]),
    "nbody_last_error": (c_char_p, [c_void_p]),
    "nbody_default_split_len": (c_int64, [c_int64]),
    "nbody_split_len": (c_int64, [c_void_p]),
    "nbody_n_total": (c_int64, [c_void_p]),
    "nbody_set_positions": (c_int, [c_void_p, c_void_p]),
    "nbody_set_velocities": (c_int, [c_void_p, c_void_p]),
    "nbody_download": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nbody_positions_device": (c_void_p, [c_void_p]),
    "nbody_velocities_device": (c_void_p, [c_void_p]),
    "nbody_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float]),
    "nbody_step_async": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float]),
    "nbody_step_n": (c_int, [c_void_p, c_int, c_float, c_float]),
    "nbody_step_n_on": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float]),
    "nbody_set_graph_replay": (c_int, [c_void_p, c_int]),
    "nbody_set_strip_len": (c_int, [c_void_p, c_int]),
    "nbody_sync": (c_int, [c_void_p]),
    "nbody_forces": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float]),
    "nbody_forces_complement": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float]),
    "nbody_update": (c_int, [c_void_p, c_void_p, c_void_p, c_float]),
    "nbody_set_integrator": (c_int, [c_void_p, c_int]),
    "nbody_invalidate_forces": (c_int, [c_void_p]),
    "nbody_kdk_prepare": (c_int, [c_void_p]),
    "nbody_kdk_kick_drift": (c_int, [c_void_p, c_void_p, c_void_p, c_float]),
    "nbody_kdk_kick": (c_int, [c_void_p, c_void_p, c_float]),
    "nbody_set_stream": (c_int, [c_void_p, c_void_p]),
    "nbody_reset_stream": (c_int, [c_void_p]),
    "nbody_energy": (c_int, [c_void_p, c_void_p, c_void_p, c_float, POINTER(c_double)]),
    "nbody_momentum": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    "nbody_timing_enable": (c_int, [c_void_p, c_int]),
    "nbody_timing_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64), POINTER(c_double),
                                  POINTER(c_int64)]),
    "nbody_timing_read_ex": (c_int, [c_void_p, POINTER(c_double)]),
    "nbody_set_force_mode": (c_int, [c_void_p, c_int]),
    "nbody_pair_once_split_len": (ctypes.c_int64, [ctypes.c_int64]),
    "nbody_sym_set_colparts": (c_int, [c_void_p, c_void_p]),
    "nbody_sym_groups": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(ctypes.c_int64)]),
    "nbody_sym_reduce": (c_int, [c_void_p]),
    "nbody_sym_rowsum": (c_int, [c_void_p]),
    "nbody_set_particle_softening": (c_int, [c_void_p, c_void_p]),
    "nbody_upload_particle_softening": (c_int, [c_void_p, c_void_p]),
    "nbody_set_rows_per_lane": (c_int, [c_void_p, c_int]),
    "nbody_set_equal_mass_path": (c_int, [c_void_p, c_int]),
    "nbody_set_early_summation": (c_int, [c_void_p, c_int]),
    "nbody_set_summation_parts": (c_int, [c_void_p, c_int]),
    "nbody_morton_order": (c_int, [c_void_p, c_int64, c_void_p]),
    "nbody_create_auto": (c_int, [POINTER(c_void_p), c_int, c_int64]),
    "nbody_force_mode": (c_int, [c_void_p]),
    "nbody_reorder": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "nbody_morton_order_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nbody_order_compute": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nbody_order_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int]),
    "nbody_order_read": (c_int, [c_void_p, c_void_p]),
    "nbody_order_identity": (c_int, [c_void_p, c_void_p, c_int64]),
    "nbody_order_set": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "nbody_order_permute_softening": (c_int, [c_void_p]),
    "nbody_multi_set_positions": (c_int, [c_void_p, c_void_p]),
    "nbody_multi_set_velocities": (c_int, [c_void_p, c_void_p]),
    "nbody_multi_timing_enable": (c_int, [c_void_p, c_int]),
    "nbody_multi_timing_read": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "nbody_multi_order": (c_int, [c_void_p, c_void_p]),
    "nbody_multi_reorder": (c_int, [c_void_p]),
    "nbody_multi_set_reorder_period": (c_int, [c_void_p, c_int64]),
    "nbody_partial_sum_bytes": (c_int64, [c_void_p]),
    "nbody_device_info": (c_int, [c_void_p, POINTER(c_int64), c_char_p, c_int]),
    # multi-GPU (csrc/nbody_multi.hip)
    "nbody_multi_geometry": (c_int, [c_int64, c_int, c_int, c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "nbody_multi_ring_schedule": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "nbody_multi_unique_id": (c_int, [c_void_p]),
    "nbody_multi_create": (c_int, [POINTER(c_void_p), c_void_p, POINTER(c_int), c_int]),
    "nbody_multi_create_rank": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_void_p]),
    "nbody_multi_destroy": (c_int, [c_void_p]),
    "nbody_multi_last_error": (c_char_p, [c_void_p]),
    "nbody_multi_set_timeout": (c_int, [c_void_p, c_double]),
    "nbody_multi_set_state": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nbody_multi_set_particle_softening": (c_int, [c_void_p, c_void_p]),
    "nbody_multi_download": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nbody_multi_step": (c_int, [c_void_p, c_float, c_float]),
    "nbody_multi_step_n": (c_int, [c_void_p, c_int, c_float, c_float]),
    "nbody_multi_step_async": (c_int, [c_void_p, c_float, c_float]),
    "nbody_multi_sync": (c_int, [c_void_p]),
    "nbody_multi_energy": (c_int, [c_void_p, c_float, POINTER(c_double)]),
    "nbody_multi_momentum": (c_int, [c_void_p, POINTER(c_double)]),
    "nbody_multi_replica_checksums": (c_int, [c_void_p, POINTER(ctypes.c_uint64)]),
    "nbody_multi_info": (c_int, [c_void_p, POINTER(c_int64)]),
    "nbody_multi_shard": (c_void_p, [c_void_p, c_int]),
    "nbody_multi_positions_device": (c_void_p, [c_void_p, c_int]),
    "nbody_multi_velocities_device": (c_void_p, [c_void_p, c_int]),
}


class MultiConfig(ctypes.Structure):
    """``nbody_multi_config`` of include/nbody.h."""
    _fields_ = [("n_bodies", c_int64), ("split_len", c_int64), ("force_mode", c_int), ("integrator", c_int),
                ("exchange", c_int), ("transport", c_int), ("body_order", c_int), ("create_timeout_s", c_int)]

_lib = None


def load() -> ctypes.CDLL:
    """Load the in-tree shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("NBODY_AMD_LIBRARY", LIB_PATH)  # an A/B build of the same ABI (tools/build_variant.sh)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  n_body_problem_amd has no CPU or PyTorch fallback.")
        try:
            # PyTorch-ROCm ships its own HIP runtime: let it load first so that this library binds to the same
            # one (loading the system runtime first and torch's second leaves the process without a device)
            import torch  # noqa: F401
        except ImportError:  # a torch-free host (ctypes only) uses the system runtime
            pass
        lib = ctypes.CDLL(path)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_names():
    return list(_PROTOTYPES)


def check(status: int, ctx=None) -> None:
    if status != NBODY_OK:
        lib = load()
        msg = lib.nbody_last_error(ctx) or b""
        text = msg.decode("utf-8", "replace") or lib.nbody_status_string(status).decode()
        raise NBodyError(status, text)
