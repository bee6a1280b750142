"""Multi-process CPU test of the sharded sort's cut logic (gloo, world_size 2 and 3).

The cuts are the product code: the library's splitter (mrs_splitter_*, the object that
mrs_dist_sort_pairs runs on the GPU box) decides from histograms that really travel between
processes (gloo all_gather), and the exchange is a real all-to-all-v received in source-rank
order.  What a CPU box cannot run, the HIP histogram / partition / sort kernels, is played by
numpy and the CPU checker here; tests/test_gpu_dist.py runs those kernels with 2-16 ranks on a
GPU.  Skewed inputs (all keys equal, one hot bucket) run with the default tolerance: no rank may
receive more than its share + 2 * tolerance.
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


def make_rank_keys(orc, rank, n_local, dist_name):
    from inputs import make_keys

    if dist_name == "hot_bucket":
        k = make_keys(orc, n_local, 1, "random", seed=rank + 1)
        hot = orc.gen_u32(rank + 50, 5, n_local) % 8 != 0
        k[hot] = np.uint32(0x7A500000) | (k[hot] & np.uint32(0xFFFFF))
        return k
    if dist_name == "all_equal":
        return np.full(n_local, 0xDEADBEEF, dtype=np.uint32)
    return make_keys(orc, n_local, 1, dist_name, seed=rank + 1)


def worker(rank, world, port, sizes, dist_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from wgpu_morton_code_radix_sort_test_amd import splitter as sp

    orc = oracle_lib.Oracle()
    n_local = sizes[rank]
    keys = make_rank_keys(orc, rank, n_local, dist_name)
    idx = np.arange(sum(sizes[:rank]), sum(sizes[:rank]) + n_local, dtype=np.uint32)
    tolerance = max(sizes) // 256
    # 1-3. histograms -> all_gather -> splitter, level by level
    s = sp.Splitter(world, 32, tolerance)
    levels = 0
    while True:
        req = s.request()
        if req is None:
            break
        shift, bits, prefixes = req
        row = torch.from_numpy(sp.local_histogram(keys, shift, bits, prefixes).astype(np.int64))
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        s.feed(np.stack([r.numpy().astype(np.uint64) for r in rows]))
        levels += 1
    res = s.result()
    # 4. stable partition by key class
    order = np.argsort(res.key_class(keys), kind="stable")
    part_k, part_v = keys[order], idx[order]
    # 5. all-to-all-v, received in source-rank order
    send = [int(c) for c in res.send_count[rank]]
    recv = res.recv_count(rank)
    outs = []
    for part in (part_k, part_v):
        out = torch.zeros(sum(recv), dtype=torch.int32)
        dist.all_to_all_single(out, torch.from_numpy(part.view(np.int32).copy()), output_split_sizes=recv, input_split_sizes=send)
        outs.append(out.numpy().view(np.uint32))
    # 6. stable local sort
    n_recv = sum(recv)
    ok, ov, _ = orc.lsd_sort(np.ascontiguousarray(outs[0]), np.ascontiguousarray(outs[1]), n_recv, 1, 1, 8, 0, 32, 256)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=ok, vals=ov, in_keys=keys, levels=levels,
             exact=res.exact_mask, tolerance=tolerance)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("dist_name", ["morton30", "few_distinct", "all_equal", "hot_bucket"])
def test_sharded_sort_gloo(tmp_path, oracle, world, dist_name):
    sizes = [5000, 3100, 4200][:world]
    port = 29500 + (os.getpid() % 2000) + world * 7 + len(dist_name)
    mp.spawn(worker, args=(world, port, sizes, dist_name, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    keys = np.concatenate([p["keys"] for p in parts])
    vals = np.concatenate([p["vals"] for p in parts])
    all_keys = np.concatenate([p["in_keys"] for p in parts])
    total = sum(sizes)
    rk, rv = oracle.stable_sort_pairs(all_keys, np.arange(total, dtype=np.uint32), total, 1, 1, 0, 32)
    # the concatenation of the ranks' slices is the globally STABLE sort
    assert np.array_equal(keys, rk)
    assert np.array_equal(vals, rv)
    # and no rank is overloaded, whatever the distribution
    tol = int(parts[0]["tolerance"])
    for r, p in enumerate(parts):
        share = total * (r + 1) // world - total * r // world
        assert p["keys"].size <= share + 2 * tol
    if dist_name == "all_equal":
        assert all(int(p["exact"]) == (1 << (world - 1)) - 1 for p in parts)
    if dist_name == "morton30":
        assert all(int(p["levels"]) <= 2 for p in parts)
