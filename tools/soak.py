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


HI_MASKS = [0xFFFFFFFF, 0x0000FFFF, 0x000000FF, 0x00000000, 0x00FF00FF, 0xFF000000, 0x000F0000, 0x0FFFFFFF]


def soak_u64(ctx, n, args):
    plan = mrs.SortPlan(ctx, n, key_words=2, payload_bytes=4, tile_items=args.tile_items)
    kb = torch.empty(2 * n, dtype=torch.int32, device=ctx.device)
    vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    bad = 0
    for it in range(args.iters):
        mask = HI_MASKS[it % len(HI_MASKS)]
        lo = ctx.generate_u32(2000 + it, 1, n)
        hi = ctx.generate_u32(2000 + it, 2, n) & (mask if mask < (1 << 31) else mask - (1 << 32))
        if it % 5 == 4:
            lo &= 0x00FFFF00  # constant low and high byte of the low word as well
        keys = torch.stack([lo, hi], dim=1).reshape(-1).contiguous()
        idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
        ref = (hi.to(torch.int64) << 32) | (lo.to(torch.int64) & 0xFFFFFFFF)
        del lo, hi
        in_b = plan.sort_pairs(keys, idx, kb, vb)
        plan.check()
        left_out = plan.passes_left_out()
        ok, ov = (kb, vb) if in_b else (keys, idx)
        o2 = ok.view(n, 2)
        k64 = (o2[:, 1].to(torch.int64) << 32) | (o2[:, 0].to(torch.int64) & 0xFFFFFFFF)
        # compare as unsigned: flip the sign bit
        ku = k64 ^ (-(1 << 63))
        sorted_ok = bool((ku[1:] >= ku[:-1]).all())
        perm_ok = bool((ref[ov.to(torch.int64)] == k64).all())
        same = k64[1:] == k64[:-1]
        stable_ok = bool((ov[1:][same] > ov[:-1][same]).all())
        if not (sorted_ok and perm_ok and stable_ok):
            bad += 1
            print(f"iteration {it}: sorted={sorted_ok} permutation={perm_ok} stable={stable_ok}", flush=True)
        if it % 8 == 7:
            print(f"{it + 1} sorts checked, {bad} bad (last mask {mask:#x}: passes left out {left_out:#04x})", flush=True)
    print("SOAK", "FAILED" if bad else "ok", f"({args.iters} sorts of 2^{args.log2n} 64-bit keys + index)")
    sys.exit(1 if bad else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--tile-items", type=int, default=0)
    ap.add_argument("--hybrid", action="store_true",
                    help="an MRS_ALGO_HYBRID plan over key bits [0, 30): two of three inputs are spread keys (the "
                         "three-move schedule) or points on surfaces (its adaptive form), the third makes the device refuse it; "
                         "sizes vary from sort to sort")
    ap.add_argument("--u64", action="store_true",
                    help="64-bit keys + index; the high word is masked differently every iteration, so passes "
                         "with a constant digit (left out on the device) come and go")
    args = ap.parse_args()
    ctx = mrs.Context(0)
    n = 1 << args.log2n
    if args.u64:
        return soak_u64(ctx, n, args)
    if args.hybrid:
        plan = mrs.SortPlan(ctx, n, end_bit=30, algo=mrs.ALGO_HYBRID)
    else:
        plan = mrs.SortPlan(ctx, n, tile_items=args.tile_items)
    cap = n
    kb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    bad = 0
    schedules = {}
    for it in range(args.iters):
        if args.hybrid:
            # sizes that move the split between the two scatters (14, 15, 16 bits) and the tile counts
            n = cap - (it * 7919 * 1021) % (cap // 2) if it % 4 else cap
        if it % 3 == 2:
            keys = ctx.generate_u32(1000 + it, 1, n)
            if args.hybrid:
                keys &= 0x3FFFFFFF
            if it % 2 == 0:
                keys &= 0xFF00FF  # few distinct digits in two passes, constant digits in the others
            idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
        else:
            xyz = ctx.generate_points(100 + it, n)
            if args.hybrid and it % 5 == 1:
                # points on a sphere shell and three planes: the fixed sub-buckets crowd and the device
                # cuts its own (the adaptive form of the schedule)
                half = n // 2
                d = xyz[:half] - 0.5
                d /= d.norm(dim=1, keepdim=True).clamp_min(1e-6)
                xyz[:half] = 0.5 + (0.45 if it % 2 else 0.3) * d
                third = (n - half) // 3
                xyz[half:half + third, 0] = 0.3
                xyz[half + third:half + 2 * third, 1] = 0.55
                xyz[half + 2 * third:, 2] = 0.8
                del d
            keys, idx = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
            del xyz
        ref = keys.clone()
        in_b = plan.sort_pairs(keys, idx, kb, vb, n=n)
        plan.check()
        if args.hybrid:
            which = plan.schedule()[0]
            schedules[which] = schedules.get(which, 0) + 1
        ok, ov = (kb[:n], vb[:n]) if in_b else (keys, idx)
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
    if args.hybrid:
        print("schedules (0 passes, 1 three moves, 2 refused by the device, 3 three moves on buckets and groups cut to the keys):", schedules)
    print("SOAK", "FAILED" if bad else "ok", f"({args.iters} sorts of up to 2^{args.log2n} pairs)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
