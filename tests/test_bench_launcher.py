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


def test_self_launch_deadline_terminates_hung_ranks():
    """a rank stuck for ever (in the real run: in a collective) must not hold the caller: after --launch-timeout the
    ranks are terminated and the launcher exits non-zero; no child survives it"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--launch-check", "--launch-timeout", "3"], {"BENCH_LAUNCH_CHECK_HANG_RANK": "1"})
    assert r.returncode != 0
    assert time.time() - t0 < 60
    assert "launch-timeout" in r.stderr
    left = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [ln for ln in left.splitlines() if "bench.py" in ln and "--launch-check" in ln and "--launch-timeout 3" in ln], left


def test_self_launch_forwards_sigterm_to_the_ranks():
    """the driver kills the launcher by pid on its own timeout: the ranks must go with it"""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_LAUNCH_CHECK_HANG_RANK"] = "0"
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--launch-timeout", "777"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(3.0)
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode != 0
    time.sleep(0.5)
    left = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [ln for ln in left.splitlines() if "bench.py" in ln and "--launch-timeout 777" in ln], left
