#!/usr/bin/env python3
"""Repeats full-size sorts and checks every result on the device (sortedness, payload is a
permutation that maps the input keys onto the output keys).  Meant to catch rare ordering bugs in
the chained scan and the ticket lists that a single run would miss."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--tile-items", type=int, default=0)
    args = ap.parse_args()
    ctx = mrs.Context(0)
    n = 1 << args.log2n
    plan = mrs.SortPlan(ctx, n, tile_items=args.tile_items)
    kb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    bad = 0
    for it in range(args.iters):
        if it % 3 == 2:
            keys = ctx.generate_u32(1000 + it, 1, n)
            if it % 2 == 0:
                keys &= 0xFF00FF  # few distinct digits in two passes, constant digits in the others
            idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
        else:
            xyz = ctx.generate_points(100 + it, n)
            keys, idx = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
            del xyz
        ref = keys.clone()
        in_b = plan.sort_pairs(keys, idx, kb, vb)
        plan.check()
        ok, ov = (kb, vb) if in_b else (keys, idx)
        k64 = ok.to(torch.int64) & 0xFFFFFFFF
        sorted_ok = bool((k64[1:] >= k64[:-1]).all())
        gathered = ref[ov.to(torch.int64)]
        perm_ok = bool((gathered == ok).all())
        # stability: equal keys keep ascending source indices
        same = k64[1:] == k64[:-1]
        stable_ok = bool((ov[1:][same] > ov[:-1][same]).all())
        if not (sorted_ok and perm_ok and stable_ok):
            bad += 1
            print(f"iteration {it}: sorted={sorted_ok} permutation={perm_ok} stable={stable_ok}", flush=True)
        if it % 10 == 9:
            print(f"{it + 1} sorts checked, {bad} bad", flush=True)
    print("SOAK", "FAILED" if bad else "ok", f"({args.iters} sorts of 2^{args.log2n} pairs)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
