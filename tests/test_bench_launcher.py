"""bench.py --gpus N started without torch.distributed.run launches its N ranks itself (the driver runs
`python3 bench.py --gpus N`).  --launch-check makes every rank report and exit before anything touches a GPU, so the
launcher runs on CPU: one JSON line from rank 0, ranks see their RANK / WORLD_SIZE, a failing rank fails the run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=120)


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                      # ONE JSON line, from rank 0
    info = json.loads(lines[0])
    assert info == {"launch_check": True, "rank": 0, "world": 2, "local_rank": 0, "master": "127.0.0.1", "gpus": 2}


def test_self_launch_propagates_failure():
    r = _run(["--gpus", "3", "--launch-check"], {"BENCH_LAUNCH_CHECK_FAIL_RANK": "2"})
    assert r.returncode != 0


def test_rank_started_by_a_launcher_does_not_spawn():
    # what torch.distributed.run does: RANK / WORLD_SIZE in the environment -> this process IS rank 1 (prints nothing)
    r = _run(["--gpus", "2", "--launch-check"], {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"})
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_single_gpu_does_not_spawn():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0
    assert json.loads(r.stdout.strip())["world"] == 1
