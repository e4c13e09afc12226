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
