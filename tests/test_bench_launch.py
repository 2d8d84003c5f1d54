"""bench.py as the driver may start it: `python bench.py --gpus N` with no launcher around it.

The parent process must bring up N ranks itself (a torch.distributed.run child, started before the parent touches the GPU)
or fail non-zero - never measure one GPU and print `n_gpus: 1`.  Covered here on the CPU over gloo with `--launch-check`
(rendezvous + rank count through the collective, nothing else); the measuring path needs GPUs.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=300)


def test_self_launch_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--launch-check"], VPC_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_env"] == 2 and out["backend"] == "gloo"


def test_refuses_to_measure_fewer_gpus_than_asked():
    import torch
    if torch.cuda.device_count() >= 64:
        return
    r = _run(["--gpus", "64", "--launch-check"])  # no rehearsal backend: 64 GPUs are not there
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_fails():
    r = _run(["--gpus", "2", "--launch-check"], WORLD_SIZE="1", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
