"""The entry point that allocates nothing: mrs_sort_pairs_noalloc against mrs_sort_pairs.

Two calls in a row through the C ABI, the second on a smaller input, each compared bit for bit
with a plan-based sort of the same data (and with the CPU checker where the size allows)."""
import numpy as np
import pytest
import torch

from gpu_util import to_dev, to_u32
from inputs import make_keys, make_vals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


def _plan_sort(ctx, mrs, keys, vals, n, begin, end):
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=begin, end_bit=end)
    ka, va = keys.clone(), vals.clone()
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    in_b = plan.sort_pairs(ka, va, kb, vb)
    plan.check()
    plan.close()
    return (kb, vb, in_b) if in_b else (ka, va, in_b)


def _noalloc_sort(ctx, keys, vals, n, begin, end):
    ka, va = keys.clone(), vals.clone()
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    in_b = ctx.sort_pairs_noalloc(ka, va, kb, vb, n, begin, end)
    # the call took no device memory (the runtime may hand back memory of objects destroyed earlier,
    # e.g. a stream of a test before this one, at any moment: free memory may grow, never shrink)
    assert torch.cuda.mem_get_info()[0] >= free_before
    assert ctx.attribute(ctx.ATTR_NOALLOC_TIMEOUT) == 0
    return (kb, vb, in_b) if in_b else (ka, va, in_b)


# the last pair fills the static scratch array completely: n = MRS_NOALLOC_MAX_N is 2048 tiles, all of its status words
@pytest.mark.parametrize("sizes", [((1 << 22) + 12345, 300007), (1 << 20, 4000), (70001, 1), (1 << 25, (1 << 20) + 1),
                                   (1 << 26, (1 << 26) - 32767)])
def test_noalloc_matches_plan_sort_twice_in_a_row(ctx, oracle, mrs, sizes):
    for j, n in enumerate(sizes):
        begin, end = (0, 32) if j == 0 else (0, 30)
        if n <= 1 << 20:
            hk = make_keys(oracle, n, 1, "morton30", seed=3 + j)
            hv = make_vals(oracle, n, 1)
            keys, vals = to_dev(hk, ctx.device), to_dev(hv, ctx.device)
        else:
            keys = ctx.generate_u32(17 + j, 2, n)
            if end == 30:
                keys &= 0x3FFFFFFF
            vals = torch.arange(n, dtype=torch.int32, device=ctx.device)
            hk = None
        want_k, want_v, want_side = _plan_sort(ctx, mrs, keys, vals, n, begin, end)
        got_k, got_v, got_side = _noalloc_sort(ctx, keys, vals, n, begin, end)
        assert got_side == want_side
        assert torch.equal(got_k[:n], want_k[:n]), (n, "keys")
        assert torch.equal(got_v[:n], want_v[:n]), (n, "payload")
        if hk is not None:
            rk, rv = oracle.stable_sort_pairs(hk, hv, n, 1, 1, begin, end)
            assert np.array_equal(to_u32(got_k)[:n], rk) and np.array_equal(to_u32(got_v)[:n], rv)


def test_noalloc_rejects_what_it_cannot_hold(ctx, mrs):
    n = ctx.NOALLOC_MAX_N + 1
    t = torch.empty(4, dtype=torch.int32, device=ctx.device)
    big = [torch.empty(n, dtype=torch.int32, device=ctx.device) for _ in range(4)]
    with pytest.raises(mrs.MrsError) as e:
        ctx.sort_pairs_noalloc(*big, n)
    assert "static scratch" in str(e.value)
    del big
    with pytest.raises(mrs.MrsError):
        ctx.sort_pairs_noalloc(t, t, t, t, 4, 5, 3)  # empty bit range
    # n = 0 is a no-op that still reports the result side
    assert ctx.sort_pairs_noalloc(t, t, t.clone(), t.clone(), 0) is False
