"""bench.py's command line, without a GPU: what its rank launcher hands to torch.distributed.run, and that it refuses to
measure a different job than the one asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forwarded_arguments_survive_the_launchers_own_parser():
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "2", "--n", "262144", "--steps", "3", "--n=4096", "--no-cpu-baseline", "--backend", "gloo"]
    out = bench.forwarded_args(argv)
    assert out == ["--gpus", "2", "--bodies", "262144", "--steps", "3", "--bodies=4096", "--no-cpu-baseline", "--backend", "gloo"]
    # the launcher's parser accepts the line and leaves the script's arguments alone
    from torch.distributed.run import get_args_parser
    ns = get_args_parser().parse_args(["--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "bench.py", *out])
    assert ns.training_script == "bench.py" and ns.training_script_args == out


def test_world_size_mismatch_and_missing_gpu_are_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                         capture_output=True, text=True)
    assert res.returncode != 0 and "WORLD_SIZE=2" in (res.stderr + res.stdout)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                                  # a multi-GPU box would run it
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True)
    assert res.returncode != 0 and not res.stdout.strip().startswith("{")      # no device here: no line, non-zero exit


def test_parent_counts_gpus_without_a_hip_runtime(monkeypatch):
    """launch_ranks must not initialise a HIP runtime in the parent (torch.cuda.device_count() can): the count
    comes from the visibility variables, else from the kfd topology, else it is left to the ranks."""
    sys.path.insert(0, ROOT)
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,5")
    assert bench.visible_gpu_count() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3")
    assert bench.visible_gpu_count() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    assert bench.visible_gpu_count() is None or bench.visible_gpu_count() >= 0     # no kfd here: None; on a GPU box: the count
    src = open(os.path.join(ROOT, "bench.py")).read()
    parent = src[src.index("def launch_ranks"):src.index("def error_line")]
    assert "device_count" not in parent and "import torch" not in parent


def test_error_line_and_per_rank_shape():
    sys.path.insert(0, ROOT)
    import argparse
    import json
    import bench
    args = argparse.Namespace(steps=20, warmup=5, n=1 << 20, transport="rccl", exchange="allgather", exchange_timeout=60.0)
    line = json.loads(bench.error_line(args, 3, 8, "nbody status -3: timed out after 60 s waiting for the step"))
    assert line["value"] is None and line["rank"] == 3 and line["n_gpus"] == 8 and "timed out" in line["error"]
    assert line["metric"] == "body-body interactions/sec" and line["config"]["exchange_timeout_s"] == 60.0
    tm = {"force_ms": 400.0, "force_launches": 40, "update_ms": 6.0, "aux_ms": 2.0, "host_enqueue_ms": 2.4,
          "pos_exchange_comm_ms": 3.0, "pos_exchange_wait_ms": 0.4, "column_sum_exchange_ms": 5.0, "reorder_ms": 0.0}
    per = bench.per_step(tm, 20)
    assert per["force_ms"] == 20.0 and per["force_launches_per_step"] == 2.0 and per["host_enqueue_ms"] == 0.12
    assert per["column_sum_exchange_ms"] == 0.25 and per["pos_exchange_wait_ms"] == 0.02
