"""Build recipe of ``libnbody_amd.so`` (HIP kernels + C ABI) for gfx950, in-tree.

``hipcc`` cross-compiles without a GPU, so this runs in the build container as well as on the
MI355X box.  The library links ``libamdhip64`` and ``librccl`` (the multi-GPU exchange of
``csrc/nbody_multi.hip``): no torch types cross the ABI (``include/nbody.h``).
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnbody_amd.so")
SOURCES = ["nbody_kernels.hip", "nbody_symmetric.hip", "nbody_order.hip", "nbody_capi.hip", "nbody_multi.hip"]
HEADERS = [os.path.join(CSRC, "nbody_kernels.h"), os.path.join(PKG_DIR, "..", "include", "nbody.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


#: per-source extra flags.  The force kernels are built without SLP vectorisation: v_pk_*_f32 buys nothing on gfx950
#: (4 cycles for two lanes' worth) and the packing moves break their hand-ordered instruction phases.
EXTRA_FLAGS = {"nbody_symmetric.hip": ["-fno-slp-vectorize"], "nbody_kernels.hip": ["-fno-slp-vectorize"]}


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the shared library if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    objdir = os.path.join(PKG_DIR, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]
    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc(), *compile_flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout + res.stderr)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed compiling {src}:\n" + res.stderr[-4000:])
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:  # one hipcc per source file
        objs = list(pool.map(compile_one, SOURCES))
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-L" + rocm_lib, "-lrccl", "-o", LIB_PATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed linking libnbody_amd.so:\n" + res.stderr[-4000:])
    return LIB_PATH


HOST_DIR = os.path.join(PKG_DIR, "..", "host")
HOST_BIN = os.path.join(HOST_DIR, "nbody_run")


def build_host(force: bool = False) -> str:
    """The thin C++ host (host/nbody_run.cpp): plain g++ against the C ABI, no HIP headers needed."""
    src = os.path.join(HOST_DIR, "nbody_run.cpp")
    deps = [src, os.path.join(HOST_DIR, "nbody_io.hpp"), os.path.join(PKG_DIR, "..", "include", "nbody.hpp"),
            os.path.join(PKG_DIR, "..", "include", "nbody.h"), LIB_PATH]
    if not force and os.path.exists(HOST_BIN) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_BIN) for d in deps):
        return HOST_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(PKG_DIR, "..", "include"), src, "-L" + PKG_DIR,
           "-lnbody_amd", "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../n_body_problem_amd", "-Wl,-rpath,/opt/rocm/lib",
           "-o", HOST_BIN]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed building host/nbody_run:\n" + res.stderr[-4000:])
    return HOST_BIN


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_host(force=True))
