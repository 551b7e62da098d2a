"""GPU suite: bench.py itself, as the driver runs it (`--multi group` at N = 1 must cost nothing)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--global-log2-batch", "23", "--steps", "8", "--warmup", "2"] + list(args),
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines                        # ONE JSON line on stdout, whatever the libraries print
    return json.loads(lines[0])


def test_group_mode_matches_the_plain_line_at_one_gpu():
    """One process per GPU (what the driver launches) against one process driving a one-member device group through the C ABI:
    same kernels, same batch, rates within 1 %; the group line says that no RCCL was needed and that its result equals the
    member's own ladder."""
    plain = run_bench()
    group = run_bench("--multi", "group")
    assert plain["config"]["parallelism"] == "shard1" and group["config"]["parallelism"] == "group1"
    assert group["config"]["rccl"]["ranks"] == 0 and group["config"]["group_parity"]["gathered_equals_each_members_own_ladder"] is True
    assert abs(group["value"] - plain["value"]) / plain["value"] < 0.01, (plain["value"], group["value"])
    for d in (plain, group):
        assert d["n_gpus"] == 1 and d["steps"] == 8 and d["roofline"]["traffic"] is not None and "parity_failures" not in d
        assert 0.6 < d["roofline"]["frac"] < 0.85
