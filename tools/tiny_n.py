import sys, time, ctypes
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import n_body_problem_amd as nb
from n_body_problem_amd import _lib
lib = _lib.load()
for n in (256, 1024, 4096):
    pos, vel = nb.plummer(n, seed=1)
    ctx = ctypes.c_void_p(None)
    assert lib.nbody_create(ctypes.byref(ctx), 0, n) == 0
    lib.nbody_set_positions(ctx, pos.ctypes.data_as(ctypes.c_void_p)); lib.nbody_set_velocities(ctx, vel.ctypes.data_as(ctypes.c_void_p))
    lib.nbody_step_n(ctx, 200, 1e-3, 1e-2)
    K = 5000
    t0 = time.perf_counter(); lib.nbody_step_n(ctx, K, 1e-3, 1e-2); wall = (time.perf_counter() - t0) / K
    lib.nbody_timing_enable(ctx, 1)
    lib.nbody_step_n(ctx, 500, 1e-3, 1e-2)
    f, u = ctypes.c_double(0), ctypes.c_double(0); fn, un = ctypes.c_int64(0), ctypes.c_int64(0)
    lib.nbody_timing_read(ctx, ctypes.byref(f), ctypes.byref(fn), ctypes.byref(u), ctypes.byref(un))
    print(f"N={n}: wall {wall*1e6:.1f} us/step; force kernel {f.value/500*1e3:.1f} us, update {u.value/500*1e3:.1f} us")
    lib.nbody_destroy(ctx)
