"""CPU: bench.py decides its launch mode before any GPU call.  `python bench.py --gpus 2` run plainly (no
WORLD_SIZE) must start its own two ranks and print exactly one JSON line; a WORLD_SIZE that disagrees with --gpus must
be rejected with a launch hint instead of an assertion after GPU initialisation."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_plain_multi_gpu_invocation_self_launches_its_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["config"]["parallelism"] == "dp2"
    assert out["max_rank_seconds"] >= 0.02          # MAX over ranks: rank 1 sleeps 20 ms


def test_four_rank_rehearsal_over_gloo():
    """the launcher path at world size 4 (rendezvous on 127.0.0.1, barrier, MAX-over-ranks timing, one JSON
    line from rank 0) rehearsed on CPU over gloo — the closest this pipeline gets to the 4- and 8-rank runs without the node."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["parallelism"] == "dp4"
    assert out["max_rank_seconds"] >= 0.04          # MAX over ranks: rank 3 sleeps 40 ms


def test_eight_rank_rehearsal_over_gloo():
    """the launch the driver uses on the 8-GPU node (`--gpus 8`, BASELINE configs 4 / 5), rehearsed on CPU
    over gloo: eight ranks rendezvous on 127.0.0.1, barrier, MAX-over-ranks timing, one JSON line from rank 0 with dp8."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak"
    assert out["max_rank_seconds"] >= 0.08          # MAX over ranks: rank 7 sleeps 80 ms


def test_single_rank_dry_run_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_world_size_mismatch_is_rejected_before_gpu_init():
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and "AssertionError" not in r.stderr
