"""bench.py --gpus N outside torch.distributed.run starts N ranks itself (no GPU needed).

The reference is a single-device program (src/main.rs:376); the 1/2/4/8-GPU curve of the metric is
this repository's, and these tests pin how its ranks are started and sized.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_selftest(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, BENCH, "--launch-selftest"] + extra, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = []
    for ln in out.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            lines.append(json.loads(ln))
    return lines


def test_gpus_2_starts_two_ranks_sharing_2pow31_pairs():
    lines = run_selftest(["--gpus", "2"])
    ranks = sorted((d for d in lines if d.get("selftest")), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1]
    assert all(d["world"] == 2 for d in ranks)
    assert [d["local_rank"] for d in ranks] == [0, 1]
    # strong scaling: the metric's curve is over 2^31 keys in all
    assert all(d["pairs_per_gpu"] == (1 << 31) // 2 and d["scaling"] == "strong" for d in ranks)
    # a rank has touched neither torch nor the library before it knows its size
    assert not any(d["torch_imported"] or d["library_imported"] for d in ranks)
    parent = [d for d in lines if d.get("selftest_parent")]
    assert len(parent) == 1 and parent[0]["children_status"] == 0
    # the process that starts the ranks never maps libmrs.so, never imports torch
    assert not parent[0]["library_imported"] and not parent[0]["torch_imported"]


def test_log2n_overrides_the_size_and_means_weak_scaling():
    lines = run_selftest(["--gpus", "2", "--log2n", "20"])
    ranks = [d for d in lines if d.get("selftest")]
    assert len(ranks) == 2 and all(d["pairs_per_gpu"] == 1 << 20 and d["scaling"] == "weak" for d in ranks)


def test_one_gpu_runs_in_process_at_the_headline_size():
    lines = run_selftest(["--gpus", "1"])
    assert len(lines) == 1 and lines[0]["world"] == 1 and lines[0]["pairs_per_gpu"] == 1 << 28


def test_under_an_outer_launcher_no_second_launch():
    # the driver's own form: torch.distributed.run has set the rank variables already
    lines = run_selftest(["--gpus", "4"], {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "4"})
    assert len(lines) == 1 and lines[0]["rank"] == 3 and lines[0]["world"] == 4
    assert lines[0]["pairs_per_gpu"] == (1 << 31) // 4


@pytest.mark.parametrize("world,expect", [(1, 1 << 28), (2, 1 << 30), (4, 1 << 29), (8, 1 << 28), (3, (1 << 31) // 3)])
def test_pairs_per_gpu(world, expect):
    sys.path.insert(0, ROOT)
    import bench

    assert bench.pairs_per_gpu(None, world) == (expect, "weak" if world == 1 else "strong")
    assert bench.pairs_per_gpu(24, world) == (1 << 24, "weak")
