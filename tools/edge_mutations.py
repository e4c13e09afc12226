#!/usr/bin/env python3
"""Can the multi-process tests see a missing hipStreamWaitEvent?  ("a build with one hipStreamWaitEvent of
nbody_multi.hip removed turns a test red".)

csrc/nbody_multi.hip numbers the event edges of its RCCL branch (EDGE(k, ...)); this tool builds, for every k, a copy of the
test-double library (tests/fake_rccl) whose nbody_multi.o was compiled with -DNB_DROP_EDGE=k -- that ONE hipStreamWaitEvent
left out -- and runs the cases of tests/test_multi_process_gpu.py against it: several processes on cuda:0, one rank each, the
state after a few steps compared bit for bit with the single-process run.  An edge counts as SEEN when at least one case
fails with it dropped (and every case passes with nothing dropped).  GPU box only.

    python tools/edge_mutations.py [edge ...]          # default: 0 (nothing dropped) 1 2 3 4 5 6
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "fake_rccl"))

EDGES = {0: "nothing dropped",
         1: "start_allgather: comm stream behind `ready` of the rank's compute stream (the rows sent are final)",
         2: "wait_allgather: consumer stream behind `done` of the comm stream (the rows have landed)",
         3: "wait_ring_hop: the force launch of chunk r - h behind the arrival of hop h",
         4: "ring: the update behind the rank's own sends (implied by edge 3 of the last hop on the RCCL branch)",
         5: "start_ring: comm stream behind `ready` of the rank's compute stream",
         6: "exchange_column_sums: comm stream behind `ready` (the rank's column sums are final)"}


def build_variant(edge: int) -> str:
    from n_body_problem_amd import build as product
    import build_fake_rccl as fake
    base = fake.build()
    if edge == 0:
        return base
    objdir = os.path.join(ROOT, "build", "obj")
    obj = os.path.join(objdir, f"nbody_multi_edge{edge}.o")
    lib = os.path.join(ROOT, "tests", "fake_rccl", f"libnbody_amd_fake_rccl_edge{edge}.so")
    deps = [os.path.join(product.CSRC, "nbody_multi.hip"), base, *product.HEADERS]
    if os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib                                   # built in the container: it travels to the GPU box with the tree
    flags = [f for f in product.FLAGS if f != "-shared"]
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in product.SOURCES if s != "nbody_multi.hip"]
    for cmd in ([product.hipcc(), *flags, f"-DNB_DROP_EDGE={edge}", "-c", os.path.join(product.CSRC, "nbody_multi.hip"), "-o", obj],
                [product.hipcc(), "-shared", "-fPIC", f"--offload-arch={product.ARCH}", *objs, obj, os.path.join(objdir, "fake_rccl.o"),
                 "-Wl,-Bsymbolic", "-lrt", "-lpthread", "-o", lib]):
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit("build failed:\n" + " ".join(cmd) + "\n" + res.stderr[-3000:])
    return lib


def main():
    if sys.argv[1:] == ["--build-only"]:
        for k in sorted(EDGES):
            print(build_variant(k))
        return
    import test_multi_process_gpu as t
    edges = [int(x) for x in sys.argv[1:]] or sorted(EDGES)
    cases = [(w, dict(c, dt=t.DT, eps=t.EPS)) for w, c in t.CASES]
    want = {}
    summary = {}
    for edge in edges:
        lib = build_variant(edge)
        results = []
        for i, (world, cfg) in enumerate(cases):
            if i not in want:
                want[i] = t.single_process(cfg, world)[:2]
            label = f"{world} ranks {cfg['force_mode']}/{cfg['exchange']}/{cfg['integrator']}/{cfg['body_order']}"
            try:
                work, codes, logs = t.run_ranks(lib, world, cfg, expect_failure=True, extra_env={"FAKE_RCCL_TIMEOUT_S": "30"})
                if any(codes):
                    results.append((label, "rank failed: " + " | ".join(x.strip().splitlines()[-1] for x in logs if x.strip())[:200]))
                    continue
                wrong = 0
                for r in range(world):
                    got = np.load(os.path.join(work, f"rank{r}.npz"))
                    wrong += not (np.array_equal(got["p"], want[i][0]) and np.array_equal(got["v"], want[i][1]))
                results.append((label, "same bits" if wrong == 0 else f"{wrong} of {world} ranks differ from the single-process run"))
            except AssertionError as e:
                results.append((label, "failed: " + str(e)[:200]))
        seen = any(r != "same bits" for _, r in results)
        summary[edge] = seen
        print(f"edge {edge}: {EDGES[edge]}\n    -> " + ("clean" if edge == 0 and not seen else "SEEN by the tests" if seen else "NOT seen"))
        for label, r in results:
            print(f"        {label}: {r}")
        sys.stdout.flush()
    print(json.dumps({"edges_seen": {str(k): bool(v) for k, v in summary.items()}}))


if __name__ == "__main__":
    main()
