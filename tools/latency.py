#!/usr/bin/env python3
"""Back-to-back latency of whole sorts at small and medium sizes (no per-launch events).

Each size sorts `--reps` fresh copies of one unsorted input on one stream and reports the mean
time per sort from two events around the whole batch.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, nargs="+", default=[12, 13, 14, 16, 18, 20, 22, 24])
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--key-words", type=int, default=1)
    ap.add_argument("--payload-bytes", type=int, default=4)
    ap.add_argument("--graph", action="store_true", help="capture one sort in a HIP graph and replay it --reps times")
    ap.add_argument("--tile-items", type=int, default=0, help="0 = the plan's choice for the size")
    args = ap.parse_args()
    ctx = mrs.Context(0)
    for l in args.log2n:
        n = 1 << l
        plan = mrs.SortPlan(ctx, n, key_words=args.key_words, payload_bytes=args.payload_bytes, tile_items=args.tile_items)
        keys = ctx.generate_u32(7, 1, n * args.key_words)
        idx = torch.arange(n * (args.payload_bytes // 4), dtype=torch.int32, device=ctx.device)
        ks = [keys.clone() for _ in range(args.reps)]
        vs = [idx.clone() for _ in range(args.reps)]
        kb, vb = torch.empty_like(keys), torch.empty_like(idx)
        plan.sort_pairs(keys.clone(), idx.clone(), kb, vb)  # warm-up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.graph:
            # the replays sort what the previous replay left (sorted keys after the first): the
            # launches and their latency are what is measured at these sizes, not the key order
            side = torch.cuda.Stream()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                plan.sort_pairs(ks[0], vs[0], kb, vb)
            graph.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                graph.replay()
            e1.record()
        else:
            e0.record()
            for k, v in zip(ks, vs):
                plan.sort_pairs(k, v, kb, vb)
            e1.record()
        torch.cuda.synchronize()
        plan.check()
        ms = e0.elapsed_time(e1) / args.reps
        print(json.dumps({"log2n": l, "ms_per_sort": round(ms, 4), "gkeys_per_s": round(n / ms / 1e6, 3),
                          "tile_items": plan.info.tile_items}), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
