"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/nbody.h
declares.  No compute is attempted without a GPU -- and the library must say so loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from n_body_problem_amd import _lib, build
    build.build_library()
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nbody.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from n_body_problem_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    assert sorted(_lib.exported_names()) == names


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_abi_version_and_status_strings(lib):
    assert lib.nbody_abi_version() == 5           # 5: non-blocking communicators (nbody_multi_config.create_timeout_s); 4: device-side body order, nbody_create_auto, NBODY_FORCE_AUTO, multi timing
    assert lib.nbody_status_string(0) == b"ok"
    for s in range(-5, 0):
        assert lib.nbody_status_string(s) not in (b"ok", b"unknown status")


def test_header_constants_match_the_python_mirror():
    """The crossover of NBODY_FORCE_AUTO and the number of summation groups are part of the ABI's contract: the host mirror must
    not drift from the header."""
    from n_body_problem_amd import system as nb
    text = open(os.path.join(ROOT, "include", "nbody.h")).read()
    defines = dict(re.findall(r"^#define\s+(NBODY_[A-Z_]+)\s+(\d+)\s*$", text, flags=re.M))
    assert int(defines["NBODY_PAIR_ONCE_MIN_BODIES"]) == nb.PAIR_ONCE_MIN_BODIES == 0     # round 4: pair-once at every size
    assert int(defines["NBODY_SYM_GROUPS"]) == 8


def test_default_split_len_is_tile_aligned_and_sharding_independent(lib):
    for n in (1, 255, 256, 257, 1024, 8192, 65536, 1 << 20, (1 << 22) + 1):
        s = lib.nbody_default_split_len(n)
        assert s % 256 == 0 and s >= 256
        assert s <= 8192 and (n + s - 1) // s <= max(128, (n + 8191) // 8192)
    assert lib.nbody_default_split_len(1 << 20) == 8192 and lib.nbody_default_split_len(1 << 22) == 8192
    assert lib.nbody_default_split_len(65536) == 512 and lib.nbody_default_split_len(10241) == 256
    # round 4, small systems (one wave = 256 rows x one split): where 256-column splits need three rounds or more on the
    # 1024 SIMDs, the split length (a multiple of 64) makes rows x splits a whole number of rounds
    assert lib.nbody_default_split_len(20225) == 320 and lib.nbody_default_split_len(20000) == 320      # 80 x 64 = 5 x 1024 waves
    assert lib.nbody_default_split_len(14791) == 256 and lib.nbody_default_split_len(28749) == 320
    for n in range(11800, 32768, 997):
        s = lib.nbody_default_split_len(n)
        waves = -(-n // s) * -(-n // 256)
        rounds = -(-waves // 1024)
        assert s % 64 == 0 and 256 <= s <= 512 and waves <= 1024 * rounds and (s == 256 or rounds >= 4)
        rb = -(-n // 256)
        assert rounds * s <= -(-(rb * rb) // 1024) * 256, (n, s)               # never more columns per SIMD than 256-column splits


def test_pair_once_split_len_depends_on_the_body_count_only(lib):
    f = lib.nbody_pair_once_split_len
    assert [f(n) for n in (0, 1, 20000, 65535, 65536, 98304, (1 << 17) - 1)] == [256, 256, 256, 256, 512, 512, 512]   # small systems: finer tiles
    assert [f(n) for n in (1 << 17, 153600, 200000, 1 << 18, (1 << 20) - 1)] == [1024] * 5   # whole passes of the eight-row loops
    assert [f(n) for n in (1 << 20, 1 << 21)] == [2048] * 2                           # half the partial sums, 2 % faster
    assert f(1 << 22) == 2048 and f(1 << 23) == 4096 and f(1 << 24) == 4096         # then bounded partial sums
    for n in (1 << 20, 1 << 21, 1 << 22):
        assert 16 * n * (n // f(n)) <= 150e9                                        # both arrays, all contexts together


def test_argument_errors_come_before_any_device_work(lib):
    ctx = ctypes.c_void_p(None)
    assert lib.nbody_create(None, 0, 16) == -1
    assert lib.nbody_create(ctypes.byref(ctx), 0, -1) == -1 and not ctx.value
    assert lib.nbody_create_shard(ctypes.byref(ctx), 0, 1024, 0, 2048, 0) == -1
    assert lib.nbody_create_shard(ctypes.byref(ctx), 0, 1024, 0, 1024, 100) == -1      # split_len % 256
    assert lib.nbody_create_shard(ctypes.byref(ctx), 0, 1024, 128, 256, 256) == -1     # row_lo % split_len
    assert b"row_lo" in lib.nbody_last_error(None)
    assert lib.nbody_step(None, None, None, None, 0.0, 0.0) == -1
    assert lib.nbody_destroy(None) == 0


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = ctypes.c_void_p(None)
    rc = lib.nbody_create(ctypes.byref(ctx), 0, 1024)
    assert rc == -4 and not ctx.value
    assert b"no CPU path" in lib.nbody_last_error(None)
    import n_body_problem_amd as nb
    with pytest.raises(nb.NBodyError):
        nb.NBodySystem(1024)
    with pytest.raises(nb.NBodyError):
        nb.step(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "n_body_problem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "nbody_oracle" not in text, f


def test_the_library_reads_one_documented_environment_variable_and_the_package_ships_no_rehearsal_harness():
    """Behaviour of a drop-in library is a function of its ABI: the only getenv left in the product's
    sources is NBODY_EXCHANGE_TIMEOUT_S (documented in include/nbody.h beside nbody_multi_set_timeout); the A/B switches of
    rounds 1-3 are gone or behind nbody_set_rows_per_lane / nbody_set_summation_parts; the torch.distributed rehearsal
    harness lives under tests/."""
    names = set()
    for sub in ("n_body_problem_amd/csrc", "include", "host"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".hip", ".h", ".hpp", ".cpp")):
                    text = open(os.path.join(dirpath, f)).read()
                    names |= set(re.findall(r'getenv\(\s*"([A-Z0-9_]+)"', text))
                    if sub != "host":
                        assert len(re.findall(r"\bgetenv\s*\(", text)) == len(re.findall(r'getenv\(\s*"NBODY_EXCHANGE_TIMEOUT_S"', text)), f
    assert "NBODY_EXCHANGE_TIMEOUT_S" in names
    assert not [n for n in names if n.startswith("NBODY_") and n not in ("NBODY_EXCHANGE_TIMEOUT_S",)], names
    assert "NBODY_EXCHANGE_TIMEOUT_S" in open(os.path.join(ROOT, "include", "nbody.h")).read()
    assert not os.path.exists(os.path.join(ROOT, "n_body_problem_amd", "sharded.py"))
    assert os.path.exists(os.path.join(ROOT, "tests", "sharded_harness.py"))


def test_header_is_plain_c_and_the_integration_patch_compiles(tmp_path):
    """include/nbody.h must be usable from C (the boundary is a C ABI: no C++ in the header), and the call-site patch of
    INTEGRATION.md section 2 -- the functions a maintainer of the reference would write -- must compile and link against the
    library as written there."""
    import subprocess
    from n_body_problem_amd import build
    build.build_library()
    src = tmp_path / "patch.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "nbody.h"
#define TIME_TICK 0.008
typedef float real;
static nbody_ctx* g_nbody = NULL;
static int m_numBodies, m_initialized;
void initialize(int numBodies) {
    m_numBodies = numBodies;
    if (nbody_create_auto(&g_nbody, /*device*/0, numBodies) != NBODY_OK) {
        fprintf(stderr, "nbody_create_auto: %s\n", nbody_last_error(NULL));
        exit(3);
    }
    m_initialized = 1;
}
void setParticlesPosition(real* data) { if (m_initialized) nbody_set_positions(g_nbody, data); }
void setParticlesVelocity(real* data) { if (m_initialized) nbody_set_velocities(g_nbody, data); }
int frame(void) {
    int rc = nbody_step(g_nbody, nbody_positions_device(g_nbody), nbody_velocities_device(g_nbody),
                        NULL, (float)TIME_TICK, 1.0e-2f);
    if (rc != NBODY_OK) printf("nbody_step: %s\n", nbody_last_error(g_nbody));
    return rc;
}
int main(void) {
    nbody_multi_config cfg = {0};
    cfg.n_bodies = 1024; cfg.force_mode = NBODY_FORCE_AUTO; cfg.body_order = NBODY_ORDER_MORTON;
    printf("abi %d, split %lld, pair-once from %d bodies, cfg %lld\n", nbody_abi_version(),
           (long long)nbody_pair_once_split_len(1 << 20), NBODY_PAIR_ONCE_MIN_BODIES, (long long)cfg.n_bodies);
    return nbody_abi_version() == NBODY_ABI_VERSION ? 0 : 1;
}
''')
    exe = tmp_path / "patch"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src),
           "-L" + os.path.join(ROOT, "n_body_problem_amd"), "-lnbody_amd", "-L/opt/rocm/lib",
           "-Wl,-rpath," + os.path.join(ROOT, "n_body_problem_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)          # no device needed: only host helpers are called
    assert run.returncode == 0 and "abi 5, split 2048, pair-once from 0 bodies" in run.stdout, run.stdout + run.stderr


def test_the_rccl_test_double_build_is_the_same_library_without_librccl():
    """tests/fake_rccl: the product's objects linked with a stand-in for the RCCL calls (test infrastructure for the
    one-rank-per-process GPU tests).  Here: it builds without a GPU, exports every symbol of the header, defines the fifteen
    RCCL entry points itself and does not depend on librccl; the product library does depend on it."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "fake_rccl"))
    import build_fake_rccl
    from n_body_problem_amd import build
    fake = ctypes.CDLL(build_fake_rccl.build())
    for name in declared_functions():
        assert hasattr(fake, name), name
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclCommCount",
                 "ncclCommGetAsyncError", "ncclGetErrorString", "ncclGetLastError", "ncclGroupStart", "ncclGroupEnd",
                 "ncclAllGather", "ncclSend", "ncclRecv", "ncclAllReduce"):
        assert hasattr(fake, name), name
    needed = lambda path: subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout  # noqa: E731
    assert "librccl" not in needed(build_fake_rccl.LIB)
    assert "librccl" in needed(build.LIB_PATH)
