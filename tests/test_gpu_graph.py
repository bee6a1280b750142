"""The sort calls under HIP graph capture: captured once, replayed on new buffer contents.

Callers with launch-bound loops (the per-frame caller, src/state.rs:109-148) replay a captured
graph instead of issuing the launches of a sort one by one; what is captured must only enqueue
work (no allocation, no synchronisation) and must read its inputs at replay time.
"""
import numpy as np
import pytest
import torch

from gpu_util import to_dev, to_u32
from inputs import make_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


def _replay_and_check(oracle, graph, fill, result, n, key_words, payload_of, end):
    for seed in (11, 12):
        keys = make_keys(oracle, n, key_words, "random", seed=seed)
        vals = payload_of(seed)
        fill(keys, vals)
        graph.replay()
        torch.cuda.synchronize()
        k, v = result()
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, 1, 0, end)
        assert np.array_equal(to_u32(k)[: n * key_words], rk)
        assert np.array_equal(to_u32(v)[:n], rv)


@pytest.mark.parametrize("n,key_words", [(3000, 1), (3000, 2), (70001, 1), (1 << 20, 1), (300007, 2)])
def test_plan_sort_replays_from_a_graph(ctx, oracle, mrs, n, key_words):
    plan = mrs.SortPlan(ctx, n, key_words=key_words, payload_bytes=4)
    ka = torch.zeros(n * key_words, dtype=torch.int32, device=ctx.device)
    va, vb = (torch.zeros(n, dtype=torch.int32, device=ctx.device) for _ in range(2))
    kb = torch.zeros_like(ka)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan.sort_pairs(ka, va, kb, vb)  # loads the kernels before anything is captured
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        in_b = plan.sort_pairs(ka, va, kb, vb)

    def fill(keys, vals):
        ka.copy_(to_dev(keys, ctx.device))
        va.copy_(to_dev(vals, ctx.device))

    rng = np.random.default_rng(5)
    _replay_and_check(oracle, graph, fill, lambda: (kb, vb) if in_b else (ka, va), n, key_words,
                      lambda seed: rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), 32 * key_words)
    plan.check()
    del graph
    plan.close()


@pytest.mark.parametrize("n", [4096, 100003, 1 << 21])
def test_indexed_and_noalloc_sorts_replay_from_a_graph(ctx, oracle, mrs, n):
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4)
    ka, va, kb, vb = (torch.zeros(n, dtype=torch.int32, device=ctx.device) for _ in range(4))
    ka2, va2, kb2, vb2 = (torch.zeros(n, dtype=torch.int32, device=ctx.device) for _ in range(4))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan.sort_pairs_indexed(ka, va, kb, vb)
        ctx.sort_pairs_noalloc(ka2, va2, kb2, vb2, n)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        in_b = plan.sort_pairs_indexed(ka, va, kb, vb)
        in_b2 = ctx.sort_pairs_noalloc(ka2, va2, kb2, vb2, n)
    iota = np.arange(n, dtype=np.uint32)
    for seed in (21, 22):
        keys = make_keys(oracle, n, 1, "random", seed=seed)
        ka.copy_(to_dev(keys, ctx.device))
        ka2.copy_(to_dev(keys, ctx.device))
        va2.copy_(to_dev(iota[::-1].copy(), ctx.device))
        graph.replay()
        torch.cuda.synchronize()
        rk, rv = oracle.stable_sort_pairs(keys, iota, n, 1, 1, 0, 32)
        assert np.array_equal(to_u32(kb if in_b else ka), rk) and np.array_equal(to_u32(vb if in_b else va), rv)
        rk, rv = oracle.stable_sort_pairs(keys, iota[::-1].copy(), n, 1, 1, 0, 32)
        assert np.array_equal(to_u32(kb2 if in_b2 else ka2), rk) and np.array_equal(to_u32(vb2 if in_b2 else va2), rv)
    plan.check()
    assert ctx.attribute(ctx.ATTR_NOALLOC_TIMEOUT) == 0
    del graph
    plan.close()
