"""Builds tests/fake_rccl/libnbody_amd_fake_rccl.so: the product's own object files (build/obj, as compiled for
libnbody_amd.so) linked with the RCCL test double of fake_rccl.cpp instead of librccl.  Test infrastructure: it lets several
processes that share one GPU run the library's one-rank-per-process path (tests/test_multi_process_gpu.py)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libnbody_amd_fake_rccl.so")
SRC = os.path.join(HERE, "fake_rccl.hip")


def build(force: bool = False) -> str:
    sys.path.insert(0, ROOT)
    from n_body_problem_amd import build as product
    objdir = os.path.join(ROOT, "build", "obj")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in product.SOURCES]
    srcs = [os.path.join(product.CSRC, s) for s in product.SOURCES] + product.HEADERS
    newest_src = max(os.path.getmtime(p) for p in srcs if os.path.exists(p))
    if not all(os.path.exists(o) and os.path.getmtime(o) >= newest_src for o in objs):
        product.build_library(force=True)          # leaves fresh objects behind
    deps = objs + [SRC]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    hipcc = product.hipcc()
    fake_obj = os.path.join(objdir, "fake_rccl.o")
    for cmd in ([hipcc, "-O2", "-std=c++17", "-fPIC", "-Wall", f"--offload-arch={product.ARCH}", "-c", SRC, "-o", fake_obj],
                [hipcc, "-shared", "-fPIC", f"--offload-arch={product.ARCH}", *objs, fake_obj, "-Wl,-Bsymbolic", "-lrt", "-lpthread",
                 "-o", LIB]):
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the RCCL test double failed:\n" + " ".join(cmd) + "\n" + res.stderr[-4000:])
    return LIB


HOST = os.path.join(HERE, "nbody_run_fake_rccl")


def build_host(force: bool = False) -> str:
    """host/nbody_run.cpp linked against the library above: the C++ host's one-process-per-GPU mode on one GPU."""
    lib = build(force)
    src = os.path.join(ROOT, "host", "nbody_run.cpp")
    deps = [src, os.path.join(ROOT, "host", "nbody_io.hpp"), os.path.join(ROOT, "include", "nbody.hpp"),
            os.path.join(ROOT, "include", "nbody.h"), lib]
    if not force and os.path.exists(HOST) and all(os.path.getmtime(d) <= os.path.getmtime(HOST) for d in deps):
        return HOST
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-L" + HERE, "-lnbody_amd_fake_rccl",
           "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-o", HOST]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed building the host against the RCCL test double:\n" + res.stderr[-4000:])
    return HOST


PROBE = os.path.join(ROOT, "tests", "rccl_probe", "rccl_nonblocking_probe")


def build_probe(force: bool = False) -> str:
    """tests/rccl_probe: the library's calling protocol on non-blocking communicators against the REAL librccl (one rank)."""
    sys.path.insert(0, ROOT)
    from n_body_problem_amd import build as product
    src = PROBE + ".hip"
    if not force and os.path.exists(PROBE) and os.path.getmtime(src) <= os.path.getmtime(PROBE):
        return PROBE
    cmd = [product.hipcc(), "-O2", "-std=c++17", "-Wno-unused-value", f"--offload-arch={product.ARCH}", src, "-L/opt/rocm/lib",
           "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", PROBE]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the RCCL protocol probe failed:\n" + res.stderr[-4000:])
    return PROBE


if __name__ == "__main__":
    print(build(force=True))
    print(build_host(force=True))
    print(build_probe(force=True))
