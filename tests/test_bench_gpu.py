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
        # algorithmic mad32 (136 per field multiplication) over the live v_mad_u64_u32 probe: 0.75 on 8 x 32-bit words
        # (rounds 1-3), 0.90-0.91 since the loop runs on nine 29-bit limbs (81 multiplies per product instead of 64 + the reduction's).
        # The floor is round 4's committed line minus 4 % box-to-box variance (0.6 would have let the ladder lose a third).
        assert FLOORS["ladder"] < d["roofline"]["frac"] < 1.0, d["roofline"]["frac"]
        # the line also says what the pipe does (tools/pipe_model.py): ~4 cycles per VALU instruction per SIMD = issue-saturated
        assert 3.7 < d["roofline"]["cycles_per_valu_instruction_per_simd"] < 4.4 and 0.9 < d["roofline"]["issue_bound_frac"] <= 1.02
        assert d["roofline"]["multiply_instructions_per_unit"] < d["roofline"]["valu_instructions_per_unit"] < d["roofline"]["algorithmic_mad32_per_unit"] * 1.3


# roofline.frac floors per workload: profiles/r04/bench_n1_*.json (r05 for the lines that are new this round) minus 4 % (boxes differ by that much in clock)
FLOORS = {"ladder": 0.86, "ladder-ref-compat": 0.63, "fixed-base": 0.86, "windowed": 0.83, "fixed-base-ct": 0.76, "brainpoolP256r1": 0.63,
          ("windowed", "brainpoolP256r1"): 0.62, ("windowed-ct", "brainpoolP256r1"): 0.58,   # the registered curve's window loop: 0.66 / 0.62 in profiles/r05
          ("fixed-base-signed", "brainpoolP256r1"): 0.58, ("fixed-base-ct", "brainpoolP256r1"): 0.56}   # ... and its signed / constant-time combs: 0.63 / 0.61


@pytest.mark.parametrize("workload,curve", [("ladder-ref-compat", "p256"), ("ladder-ref-compat", "secp256k1"), ("fixed-base", "p256"), ("windowed", "p256"), ("fixed-base-ct", "p256"), ("ladder", "brainpoolP256r1"),
                                            ("windowed", "brainpoolP256r1"), ("windowed-ct", "brainpoolP256r1"), ("fixed-base-signed", "brainpoolP256r1"), ("fixed-base-ct", "brainpoolP256r1")])
def test_every_workload_stays_at_its_committed_fraction_of_the_roof(workload, curve):
    """a regression guard per workload, not only for the headline -- the reference-compatible ladder (the mode that is identical to
    the reference on every lane), BASELINE configs[2]'s LDS comb, the windowed variable-base path, the constant-time comb, and a curve registered at run time (its ladder and its window loop, plain and constant-time)."""
    d = run_bench("--workload", workload, "--curve", curve, "--global-log2-batch", "22")
    floor = FLOORS.get((workload, curve)) or (FLOORS["brainpoolP256r1"] if curve == "brainpoolP256r1" else FLOORS[workload])
    assert floor < d["roofline"]["frac"] < 1.0, (workload, curve, d["roofline"]["frac"])
    assert "parity_failures" not in d


def test_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` as the driver may start it (no torch.distributed.run around it): the launcher starts two
    ranks; with ECSIMD_BENCH_REHEARSE_ONE_GPU=1 both use cuda:0 and gather over gloo (RCCL refuses two ranks on one device),
    so everything an N > 1 run does except RCCL itself executes on this one-GPU box: the strong-scaling shard plan, rank 1's
    shard arriving at rank 0 and being checked against rank 0's own ladder, the device-group leg in a child process (two
    members on one device) while the ranks wait on the host group, one JSON line, exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["ECSIMD_BENCH_REHEARSE_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--global-log2-batch", "21", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and "rehearsal" in d and "parity_failures" not in d
    assert d["config"]["global_batch"] == 1 << 21 and d["config"]["per_gpu_batch"] == 1 << 20 and d["config"]["parallelism"] == "shard2+rccl_gather"
    g = d["config"]["gather"]
    assert g["sample_check"] == {"lanes_per_rank": 1024, "ranks_checked": 1, "ranks_differing": []}
    assert g["bytes_per_rank_per_step"] == 3 * (1 << 20) * 32 and g["value_compute_only"] > 0
    mg = d["multi_group"]
    assert mg["ok"] is True, mg
    assert mg["parity"] == {"gathered_equals_each_members_own_ladder": True, "members": 2}
    assert d["roofline"]["frac"] > 0.2            # two ranks share the card: each ladder runs at about half rate
