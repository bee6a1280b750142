#!/usr/bin/env python3
"""Checks MRS_ALGO_HYBRID plans against torch's stable sort on the device, both schedules.

Sizes the CPU checker would take minutes for are covered here: the expected result of a stable
sort of (key, index) pairs is unique, so any stable sort serves as the second opinion.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def expected(keys, begin_bit, end_bit):
    k = keys.to(torch.int64) & 0xFFFFFFFF
    field = (k >> begin_bit) & ((1 << (end_bit - begin_bit)) - 1)
    order = torch.sort(field, stable=True)[1]
    return keys[order], order.to(torch.int32)


def one(ctx, n, dist, begin_bit, end_bit, indexed=False):
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=begin_bit, end_bit=end_bit,
                        algo=mrs.ALGO_HYBRID)
    if dist == "morton30":
        xyz = ctx.generate_points(11, n)
        keys, _ = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
    elif dist == "random":
        keys = ctx.generate_u32(5, 2, n)
    elif dist == "equal":
        keys = torch.full((n,), 0x2ABCDEF1, dtype=torch.int32, device=ctx.device)
    elif dist == "fewvalues":
        keys = (ctx.generate_u32(5, 3, n) & 3) * 0x01010101
    elif dist == "onebucket":  # one first-level bucket, spread below it
        keys = (ctx.generate_u32(5, 4, n) & 0x003FFFFF) | (0x15 << 22)
    else:
        raise ValueError(dist)
    idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
    ek, ev = expected(keys, begin_bit, end_bit)
    ka, va = keys.clone(), (torch.empty_like(idx) if indexed else idx.clone())
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    in_b = plan.sort_pairs_indexed(ka, va, kb, vb) if indexed else plan.sort_pairs(ka, va, kb, vb)
    plan.check()
    ok_side = in_b == (plan.info.num_passes % 2 == 1)
    rk, rv = (kb, vb) if in_b else (ka, va)
    ok = ok_side and torch.equal(rk, ek) and torch.equal(rv, ev)
    print("%-10s n=%-10d bits [%d,%d) indexed=%d %s" % (dist, n, begin_bit, end_bit, indexed, "ok" if ok else "MISMATCH"),
          flush=True)
    if not ok:
        bad = (rk != ek).nonzero()
        print("  first key mismatch at", bad[0].item() if bad.numel() else None, "of", bad.numel(),
              "; payload mismatches", (rv != ev).sum().item())
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also 2^26 and 2^28 pairs")
    ap.add_argument("--sizes", type=int, nargs="*", default=[],
                    help="only these element counts: 30-bit codes on bits [0, 30), random keys on [0, 32) and [0, 28)")
    args = ap.parse_args()
    ctx = mrs.Context(0)
    ok = True
    if args.sizes:
        for n in args.sizes:
            ok &= one(ctx, n, "morton30", 0, 30)
            torch.cuda.empty_cache()
            ok &= one(ctx, n, "random", 0, 32, indexed=True)
            torch.cuda.empty_cache()
            ok &= one(ctx, n, "random", 0, 28)
            torch.cuda.empty_cache()
        print("ALL OK" if ok else "FAILURES")
        return 0 if ok else 1
    sizes = [4097, 5000, 65536, 100003, (1 << 20) + 123, (1 << 22) + 7]
    for n in sizes:
        for dist in ("morton30", "random", "equal", "fewvalues", "onebucket"):
            bits = (0, 30) if dist == "morton30" else (0, 32)
            ok &= one(ctx, n, dist, *bits)
    ok &= one(ctx, 300000, "random", 3, 29)
    ok &= one(ctx, 300000, "random", 0, 25)
    ok &= one(ctx, 300000, "morton30", 0, 30, indexed=True)
    if args.big:
        for n in (1 << 26, 1 << 28):
            ok &= one(ctx, n, "morton30", 0, 30)
            ok &= one(ctx, n, "random", 0, 32)
        ok &= one(ctx, 1 << 26, "equal", 0, 32)
        ok &= one(ctx, 1 << 26, "onebucket", 0, 32)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
