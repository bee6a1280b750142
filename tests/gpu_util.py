"""Helpers for the GPU parity tests: numpy <-> device tensors, calls through the C ABI."""
import numpy as np
import torch


def to_dev(arr, device):
    """numpy u32/u8/f32 array -> device tensor with the same bytes."""
    if arr is None:
        return None
    if arr.dtype == np.uint32:
        return torch.from_numpy(arr.view(np.int32).copy()).to(device)
    return torch.from_numpy(arr.copy()).to(device)


def to_u32(t):
    return t.cpu().numpy().view(np.uint32)


def gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, bits, begin, end, algo, tile_items=0,
             capacity=None, indexed=False):
    """Runs mrs_sort_pairs (or, with `indexed`, mrs_sort_pairs_indexed on a payload buffer filled
    with junk); returns (keys_out, vals_out, result_in_b) as numpy u32 arrays."""
    dev = ctx.device
    plan = mrs.SortPlan(ctx, capacity or max(n, 1), key_words=key_words, payload_bytes=4 * payload_words,
                        digit_bits=bits, begin_bit=begin, end_bit=end, algo=algo, tile_items=tile_items)
    ka = to_dev(keys, dev) if n else torch.zeros(key_words, dtype=torch.int32, device=dev)
    kb = torch.full_like(ka, -1)
    va = vb = None
    if payload_words:
        va = to_dev(vals, dev) if n else torch.zeros(payload_words, dtype=torch.int32, device=dev)
        vb = torch.full_like(va, -1)
    if indexed:
        va.fill_(0x3C3C3C3C)
        in_b = plan.sort_pairs_indexed(ka, va, kb, vb, n=n)
    else:
        in_b = plan.sort_pairs(ka, va, kb, vb, n=n)
    plan.check()
    ko = to_u32(kb if in_b else ka)[: n * key_words]
    vo = to_u32(vb if in_b else va)[: n * payload_words] if payload_words else None
    plan.close()
    return ko, vo, in_b
