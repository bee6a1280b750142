#!/usr/bin/env python3
"""Runs a few sorts of 2^log2n Morton (key, index) pairs and prints per-launch event times.

A small driver for profiling (rocprofv3 --kernel-trace / --pmc) and A/B tuning.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def surface_points(xyz, dist):
    """Turns uniform points (n x 3, in place) into the named distribution."""
    n = xyz.shape[0]
    if dist == "cluster5":
        m = n // 20
        xyz[:m] = xyz[:m] * 0.02 + 0.4
    elif dist == "shell":
        d = xyz - 0.5
        d /= d.norm(dim=1, keepdim=True).clamp_min(1e-6)
        xyz.copy_(0.5 + 0.45 * d)
    elif dist == "planes":
        third = n // 3
        xyz[:third, 0] = 0.3
        xyz[third:2 * third, 1] = 0.55
        xyz[2 * third:, 2] = 0.8
    elif dist == "surface":  # bench.py's "surface" input: half shell, half planes
        half = n // 2
        d = xyz[:half] - 0.5
        d /= d.norm(dim=1, keepdim=True).clamp_min(1e-6)
        xyz[:half] = 0.5 + 0.45 * d
        third = (n - half) // 3
        xyz[half:half + third, 0] = 0.3
        xyz[half + third:half + 2 * third, 1] = 0.55
        xyz[half + 2 * third:, 2] = 0.8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--n", type=int, default=0, help="element count (overrides --log2n)")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--tile-items", type=int, default=0)
    ap.add_argument("--digit-bits", type=int, default=8)
    ap.add_argument("--algo", default="onesweep")
    ap.add_argument("--key-words", type=int, default=1)
    ap.add_argument("--end-bit", type=int, default=0, help="one past the last key bit sorted on (0: the whole key)")
    ap.add_argument("--payload-bytes", type=int, default=4)
    ap.add_argument("--dist", default="morton30", choices=["morton30", "random", "sorted", "equal", "cluster5", "shell", "planes", "dup64", "surface"],
                    help="morton30: codes of uniform points; cluster5: the same with 5 %% of the points in a small cube; "
                         "shell / planes: codes of points on a sphere shell / on three axis-aligned planes")
    ap.add_argument("--leading-bits", type=int, default=0, help="MRS_ATTR_TEST_HYBRID_LEADING_BITS (14..16; 0 = the library's choice)")
    ap.add_argument("--adaptive", default="auto", choices=["auto", "force", "off"],
                    help="hybrid plans: the cut that follows the keys' density is the device's choice, forced, or not offered")
    ap.add_argument("--pad-bytes", type=int, default=-1,
                    help="32-bit keys + index: the four buffers (keys, index, and the ping-pong pair) are carved out of "
                         "one allocation, each this many bytes further from a multiple of the buffer size than the one "
                         "before (-1: four allocations of their own)")
    ap.add_argument("--hi-mask", type=lambda v: int(v, 0), default=0xFFFFFFFF,
                    help="64-bit keys: mask applied to the high word (short codes have constant high bytes)")
    ap.add_argument("--gather", action="store_true",
                    help="sort (key, source index) and move each record of --payload-bytes once (mrs_sort_gather)")
    args = ap.parse_args()
    ctx = mrs.Context(0)
    if args.leading_bits:
        ctx.set_attribute(ctx.ATTR_TEST_HYBRID_LEADING_BITS, args.leading_bits)
    if args.adaptive == "force":
        ctx.set_attribute(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    elif args.adaptive == "off":
        ctx.set_attribute(ctx.ATTR_TEST_HYBRID_NO_ADAPTIVE, 1)
    n = args.n or (1 << args.log2n)
    algo = {"onesweep": mrs.ALGO_ONESWEEP, "hybrid": mrs.ALGO_HYBRID}.get(args.algo, mrs.ALGO_THREE_KERNEL)
    plan = mrs.SortPlan(ctx, n, key_words=args.key_words, payload_bytes=4 if args.gather else args.payload_bytes,
                        digit_bits=args.digit_bits, algo=algo, tile_items=args.tile_items,
                        end_bit=args.end_bit or None)
    plan.set_timing(True)
    kw, pw = args.key_words, args.payload_bytes // 4
    arena = None
    if args.pad_bytes >= 0 and kw == 1 and pw == 1 and not args.gather:
        step = n + args.pad_bytes // 4
        arena = torch.empty(4 * step, dtype=torch.int32, device=ctx.device)
        kb, vb = arena[2 * step:2 * step + n], arena[3 * step:3 * step + n]
    else:
        kb = torch.empty(n * kw, dtype=torch.int32, device=ctx.device)
        vb = torch.empty(max(1, n * pw), dtype=torch.int32, device=ctx.device)
    rows = []
    for it in range(args.iters):
        if args.dist in ("morton30", "cluster5", "shell", "planes", "dup64", "surface"):
            xyz = ctx.generate_points(7 + it, n)
            mn, mx = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
            surface_points(xyz, args.dist)
            keys, idx = ctx.morton_encode30(xyz, mn, mx)
            del xyz
            if args.dist == "dup64":  # evenly spread codes, 64 copies of each on average
                keys &= ~0x3F
            if kw == 2:
                hi = ctx.generate_u32(3, it, n)
                if args.hi_mask != 0xFFFFFFFF:
                    hi &= args.hi_mask if args.hi_mask < (1 << 31) else args.hi_mask - (1 << 32)
                keys = torch.stack([keys, hi], dim=1).reshape(-1).contiguous()
        else:
            keys = ctx.generate_u32(7 + it, 1, n * kw)
            if args.dist == "sorted":
                keys = torch.sort(keys)[0]
            elif args.dist == "equal":
                keys.fill_(12345)
            idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
        vals = idx if pw == 1 else (ctx.generate_u32(5, it, n * pw) if pw else None)
        if arena is not None:
            step = n + args.pad_bytes // 4
            arena[0:n].copy_(keys)
            arena[step:step + n].copy_(vals)
            del keys, idx, vals
            keys, vals = arena[0:n], arena[step:step + n]
            idx = vals
        if args.gather:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.sort_gather(keys, kb, vals, vb, args.payload_bytes)
            e1.record()
            torch.cuda.synchronize()
            tm = plan.timings() + [("whole call (events around it)", e0.elapsed_time(e1))]
        else:
            plan.sort_pairs(keys, vals, kb, vb if pw else None)
            tm = plan.timings()
        plan.check()
        rows.append(tm)
    last = rows[-1]
    total = sum(ms for lab, ms in last if not lab.startswith("whole call"))
    extra = {}
    if algo == mrs.ALGO_HYBRID:
        schedule, largest = plan.schedule()
        subs, pairs, cap = plan.oversized()
        extra = {"schedule": {0: "passes", 1: "msd", 2: "msd refused by the device", 3: "msd adaptive"}[schedule], "largest_sub_bucket": largest,
                 "oversized_sub_buckets": subs, "oversized_pairs": pairs, "workgroup_capacity": cap}
        if schedule == 3:
            extra["adaptive_buckets"], extra["adaptive_groups"] = plan.adaptive()
    print(json.dumps({"n": n, "dist": args.dist, "tile_items": plan.info.tile_items, "threads": plan.info.threads,
                      "launches": [(lab, round(ms, 4)) for lab, ms in last], "total_ms": round(total, 4),
                      "gkeys_per_s": round(n / total / 1e6, 3), **extra}))


if __name__ == "__main__":
    main()
