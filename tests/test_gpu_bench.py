"""GPU suite (-m gpu): the benchmark's N > 1 path end to end on real kernels.  No multi-GPU node is needed: with
BENCH_BACKEND=gloo the two ranks share the box's GPU and the gather of the decoded bytes goes through host memory --
everything else (self-launch of fresh rank processes, sharded decode on the HIP kernels, sharding.gather_decoded,
barrier + max-over-ranks timing, ONE JSON line from rank 0) is what the driver's `python3 bench.py --gpus N` runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("config", ["cfg3", "cfg2"])
def test_two_ranks_decode_and_gather(built_library, config):
    common = ["--config", config, "--log2-batch", "14", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc-traffic"]
    one = _bench(["--gpus", "1"] + common)
    two = _bench(["--gpus", "2", "--launch-timeout", "300"] + common, {"BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak"
    assert two["config"]["codewords_per_gpu"] == 1 << 14
    assert two["metric"] == one["metric"] and two["unit"] == "codewords/s"
    # rank 0 decodes the same rows in both runs (the seed depends on the rank only): identical payload statistics
    assert two["payload_match_fraction"] == one["payload_match_fraction"]
    assert two["detector_pass_fraction"] == one["detector_pass_fraction"]
    assert two["value"] > 0 and one["roofline"]["kernel"] == two["roofline"]["kernel"]


def test_default_line_carries_roofline_and_issue(built_library):
    """the line the driver records: roofline (achieved / peak / frac / traffic) and, with the profiler on the box, the
    instruction-issue mix of the dominant kernel"""
    import shutil
    res = _bench(["--gpus", "1", "--config", "cfg3", "--log2-batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["kernel"] in ("scl_lane_kernel", "scl_decode_kernel")
    if shutil.which("rocprofv3"):
        assert rf["traffic"] and rf["traffic"] > 0
        iss = rf["issue"]
        assert 0 < iss["salu_frac"] < 1 and 0 < iss["valu_frac"] < 1 and 0 <= iss["wait_frac"] < 1
