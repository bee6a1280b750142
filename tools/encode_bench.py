#!/usr/bin/env python3
"""Times mrs_morton_encode30 (20 algorithmic bytes per point) at 2^log2n points."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    ctx = mrs.Context(0)
    n = 1 << args.log2n
    xyz = ctx.generate_points(7, n)
    keys, idx = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1], keys, idx)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(json.dumps({"points": n, "ms": round(ms, 4), "TBps_20B": round(20.0 * n / (ms * 1e-3) / 1e12, 3)}))


if __name__ == "__main__":
    main()
