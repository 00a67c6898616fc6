"""`python bench.py --gpus 2` end to end on the one-GPU box: no launcher, the two ranks share cuda:0 over gloo
(GNM_BENCH_SHARE_GPU=1, the bench's test hook) -- everything of the N > 1 path except RCCL itself: self-launch,
per-rank pools, hipGraph replay or eager, the flat-gradient all-reduce, max-over-ranks timing, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["GNM_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "32", "--pool", "64",
                        "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["config"]["graphs_per_gpu"] == 32
    assert out["value"] > 0 and out["ms_per_step"] < 200, out["ms_per_step"]     # round 1: 279-404 ms through gloo
    # the object the first real `bench.py --gpus 8` line will carry: who took part, what the
    # collective was, how the launch mode was chosen, and that the ranks agree where they must
    cc = out["collective"]
    assert cc["world"] == 2 and cc["backend"] == "gloo" and cc["forced_single_rank_test_hook"] is False
    assert cc["op"].startswith("gloo staging"), cc
    assert set(cc["launch_mode_trial_ms"]) >= {"graph", "eager"} and cc["launch_mode"] in cc["launch_mode_trial_ms"], cc
    rc = cc["rank_check"]
    assert rc["params_agree_after_broadcast"] is True and rc["grads_agree_after_allreduce"] is True, rc
    assert rc["grad_abs_sum"] > 0 and len(rc["final_loss_per_rank"]) == 2
    assert all(v == v and abs(v) < 1e6 for v in rc["final_loss_per_rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--neighbor-pooling", "max"], ["--config", "c4", "--batch", "8", "--pool", "8"],
                                   ["--graph", "off"], ["--no-learn-eps", "--graph-pooling", "average"]])
def test_bench_other_configurations_print_one_line(extra):
    """the non-default bench configurations (DESIGN.md section 6's matrix) keep working: one JSON line, a finite loss"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    if "--batch" not in extra:
        cmd += ["--batch", "16", "--pool", "16"]
    r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert out["final_loss"] == out["final_loss"] and abs(out["final_loss"]) < 1e6      # finite


@pytest.mark.gpu
def test_bench_rccl_path_at_world_size_one():
    """The part of the N > 1 path a one-GPU box CAN run on RCCL itself (two ranks may not share a GPU under RCCL):
    GNM_BENCH_FORCE_DIST=1 initialises the "nccl" process group at world size 1 and runs the multi-rank step path on
    it -- communicator init, the AVG all-reduce of the flat gradient buffer (gnm/parallel.py), its capture inside the
    step's hipGraph, the launch-mode trial.  Lines that had never executed before round 3."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["GNM_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "32", "--pool", "64",
                        "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    cc = out["collective"]
    assert cc["backend"] == "nccl" and cc["forced_single_rank_test_hook"] is True
    assert cc["op"].startswith(("AVG", "SUM")), cc
    assert set(cc["launch_mode_trial_ms"]) >= {"graph", "eager"}, cc
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]
    print("collective:", json.dumps(cc))
