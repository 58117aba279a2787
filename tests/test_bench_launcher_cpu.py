"""`python bench.py --gpus N` with no torchrun around it must start its own N ranks (the reference's
tools/batch_eval.py:80-95 launches its per-GPU workers itself) and still print ONE JSON line with per-rank rates.  The
step is a host stub on gloo ranks: this tests the launcher and the measurement protocol, not the model."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-step", "--steps", "3", "--warmup", "0"]
                          + extra, capture_output=True, text=True, timeout=300, env=env)


def test_bench_spawns_its_own_ranks():
    cp = _run(["--gpus", "2"])
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and len(d["config"]["per_rank_images_per_sec"]) == 2
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["rccl_world_size"] == 2 and d["config"]["backend"] == "gloo"    # the backend saw both ranks


def test_bench_eight_ranks_with_an_empty_rank():
    """The 8-GPU launch the driver's SCALE run makes, as far as a GPU-less container can rehearse it: eight gloo ranks on one
    host, rank 5 produces no detection at all (its padded all_gather slot is empty), rows still arrive in rank order."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["CSAM_BENCH_STUB_EMPTY_RANK"] = "5"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-step", "--gpus", "8", "--steps", "3",
                         "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and len(d["config"]["per_rank_images_per_sec"]) == 8
    assert d["config"]["rccl_world_size"] == 8 and d["config"]["backend"] == "gloo"


def test_bench_single_rank_needs_no_launcher():
    cp = _run(["--gpus", "1"])
    assert cp.returncode == 0, cp.stderr[-2000:]
    d = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and len(d["config"]["per_rank_images_per_sec"]) == 1


def test_bench_rank_failure_propagates():
    """a rank that dies must make the launcher exit non-zero instead of hanging the others in a collective"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CSAM_BENCH_STUB_FAIL_RANK"] = "1"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-step", "--gpus", "2", "--steps", "2",
                         "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode != 0
