"""Multi-process CPU test of the distributed sort's orchestration (gloo, world_size 2 and 3).

The collectives and the bucket -> rank logic are the product code; the local stages
(histogram, partition, sort) are replaced by a checker-backed double so that the test
runs without a GPU.  The HIP local stages are covered by the GPU parity tests.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class CpuLocalOps:
    """Test double for HipLocalOps: same interface, numpy + the CPU oracle underneath."""

    def __init__(self, oracle, key_words=1, payload_words=1):
        self.orc = oracle
        self.kw, self.pw = key_words, payload_words

    def empty(self, n, words):
        return torch.zeros(max(1, n * words), dtype=torch.int32)

    def bucket_histogram(self, keys, n, shift, bits):
        k = keys.numpy().view(np.uint32)[:n]
        return torch.from_numpy(np.bincount((k >> np.uint32(shift)) & np.uint32((1 << bits) - 1),
                                            minlength=1 << bits).astype(np.int64))

    def partition(self, keys, vals, n, keys_out, vals_out, shift, bits, table, parts):
        k = keys.numpy().view(np.uint32)[:n]
        v = vals.numpy().view(np.uint32)[:n]
        dest = table[(k >> np.uint32(shift)) & np.uint32((1 << bits) - 1)]
        order = np.argsort(dest, kind="stable")
        keys_out.numpy().view(np.uint32)[:n] = k[order]
        vals_out.numpy().view(np.uint32)[:n] = v[order]

    def sort(self, keys_a, vals_a, keys_b, vals_b, n):
        k = np.ascontiguousarray(keys_a.numpy().view(np.uint32)[:n])
        v = np.ascontiguousarray(vals_a.numpy().view(np.uint32)[:n])
        ko, vo, _ = self.orc.lsd_sort(k, v, n, 1, 1, 8, 0, 32, 256)
        keys_b.numpy().view(np.uint32)[:n] = ko
        vals_b.numpy().view(np.uint32)[:n] = vo
        return keys_b, vals_b

    def mark(self):
        import time
        return time.perf_counter()

    @staticmethod
    def elapsed_ms(a, b):
        return (b - a) * 1e3


def worker(rank, world, port, n_local, dist_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from inputs import make_keys
    from wgpu_morton_code_radix_sort_test_amd.dist import DistributedSorter

    orc = oracle_lib.Oracle()
    all_keys = make_keys(orc, n_local * world, 1, dist_name)
    keys = all_keys[rank * n_local:(rank + 1) * n_local].copy()
    idx = np.arange(rank * n_local, (rank + 1) * n_local, dtype=np.uint32)
    sorter = DistributedSorter(None, n_local, world, rank, ops=CpuLocalOps(orc), slack=float(world) + 0.5)
    ok, ov = sorter.sort(torch.from_numpy(keys.view(np.int32)), torch.from_numpy(idx.view(np.int32)))
    good = sorter.verify(ok, ov)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=ok.numpy().view(np.uint32), vals=ov.numpy().view(np.uint32),
             good=good, breakdown=len(sorter.last_breakdown()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("dist_name", ["morton30", "few_distinct", "all_equal"])
def test_distributed_sort_gloo(tmp_path, oracle, world, dist_name):
    from inputs import make_keys

    n_local = 5000
    port = 29500 + (os.getpid() % 2000) + world * 7 + len(dist_name)
    mp.spawn(worker, args=(world, port, n_local, dist_name, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert all(bool(p["good"]) for p in parts)
    assert all(int(p["breakdown"]) == 5 for p in parts)
    keys = np.concatenate([p["keys"] for p in parts])
    vals = np.concatenate([p["vals"] for p in parts])
    all_keys = make_keys(oracle, n_local * world, 1, dist_name)
    all_idx = np.arange(n_local * world, dtype=np.uint32)
    rk, rv = oracle.stable_sort_pairs(all_keys, all_idx, n_local * world, 1, 1, 0, 32)
    # the concatenation of the ranks' slices is the globally STABLE sort
    assert np.array_equal(keys, rk)
    assert np.array_equal(vals, rv)
