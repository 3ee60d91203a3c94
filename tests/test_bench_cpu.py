"""CPU: bench.py decides its launch mode before any GPU call.  `python bench.py --gpus 2` run plainly (no
WORLD_SIZE) must start its own two ranks and print exactly one JSON line; a WORLD_SIZE that disagrees with --gpus must
be rejected with a launch hint instead of an assertion after GPU initialisation."""
import json
import os
import subprocess
import sys

import pytest

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
    # the ranks' shards of the global batch carry (nearly) the same number of objects: the weak-scaling straggler term
    objs = out["config"]["objects_per_rank"]
    assert len(objs) == 8 and (max(objs) - min(objs)) <= 0.02 * max(objs), objs


def test_object_balanced_sharding_of_the_global_batch():
    """agl.synth.balanced_object_counts: every rank gets the same number of images, the ranks' object counts differ by <= 2 %, the global
    multiset of object counts is the plain draw's, and the table is a pure function of (images per rank, world, seed)."""
    sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
    import numpy as np
    from agl import synth
    for world, per in ((2, 64), (4, 64), (8, 64), (8, 32), (3, 10)):
        rows = synth.balanced_object_counts(per, world, seed=1234)
        assert [len(r) for r in rows] == [per] * world
        sums = [int(r.sum()) for r in rows]
        assert max(sums) - min(sums) <= max(1, 0.02 * max(sums)), (world, per, sums)
        plain = np.random.default_rng(1234).integers(3, 10, size=per * world)
        assert sorted(np.concatenate(rows).tolist()) == sorted(plain.tolist())
        again = synth.balanced_object_counts(per, world, seed=1234)
        assert all((a == b).all() for a, b in zip(rows, again))


def test_committed_counter_profile_matches_the_library():
    """bench.py quotes HBM traffic from profiles/r05_hbm_traffic.json (rocprofv3 PMC passes).  The figure describes the kernels it was
    collected on: the per-kernel table it points to must name kernels that exist in the built library (demangled symbols of libagl.so),
    and the file must carry the ABI version and split form it was collected for (bench.py emits null for any other library)."""
    import csv, re
    tj = os.path.join(ROOT, "profiles", "r05_hbm_traffic.json")
    if not os.path.exists(tj):
        pytest.skip("no counter profile committed yet")
    data = json.load(open(tj))
    lib = os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd", "agl", "libagl.so")
    syms = subprocess.run("nm %s | c++filt" % lib, shell=True, capture_output=True, text=True).stdout
    have = set(re.findall(r"(?:\(anonymous namespace\)::)?(\w+_k|igemm_f32|patch_conv|small_cout_conv|splitk_epilogue|slab_reduce\w*|bn_stats_\w+|norm_\w+|lstm_gates_\w+)\b", syms))
    header = open(os.path.join(ROOT, "include", "agl.h")).read()
    abi = int(re.search(r"#define AGL_ABI_VERSION (\d+)", header).group(1))
    for tag, d in data.items():
        assert d.get("abi") == abi and d.get("split_products") in (3, 6) and d.get("collected_at"), (tag, d)
        rows = list(csv.reader(open(os.path.join(ROOT, d["source"]))))[1:]
        names = {re.sub(r"^void ", "", r[0]).split("<")[0].split("(")[0] for r in rows[:40]}
        ours = {n for n in names if not n.startswith("at::") and not n.startswith("void at::") and "Cijk" not in n and "rccl" not in n.lower() and not n.startswith("__amd_rocclr")}      # (runtime blit kernels are not the library's)
        missing = {n for n in ours if n not in have}
        assert not missing, (tag, "kernels of the committed profile that the library no longer has", sorted(missing))


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
