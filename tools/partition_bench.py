#!/usr/bin/env python3
"""Times mrs_bucket_histogram and mrs_partition (the local stages of the multi-GPU split)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def bucket_to_part_table(counts, parts):
    """u8 table: bucket b goes to the part its first element falls in when the total is cut evenly."""
    c = np.asarray(counts, dtype=np.int64)
    before = np.concatenate([[0], np.cumsum(c)[:-1]])
    return np.minimum(parts - 1, before * parts // max(1, int(c.sum()))).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    args = ap.parse_args()
    ctx = mrs.Context(0)
    n = 1 << args.log2n
    xyz = ctx.generate_points(7, n)
    keys, idx = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
    del xyz
    plan = mrs.SortPlan(ctx, n)
    ko, vo = torch.empty_like(keys), torch.empty_like(idx)
    for parts in (1, 2, 4, 8, 16):
        counts = ctx.bucket_histogram(keys, n, 1, 20, 12)
        table = bucket_to_part_table(counts.cpu().numpy(), parts)
        table_dev = torch.from_numpy(table).to(ctx.device)
        for what in ("bucket_histogram", "partition"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(4):
                if rep == 1:
                    e0.record()
                if what == "partition":
                    plan.partition(keys, idx, ko, vo, 20, 12, table_dev, parts, n=n)
                else:
                    ctx.bucket_histogram(keys, n, 1, 20, 12)
            e1.record()
            torch.cuda.synchronize()
            print(f"parts {parts:2d} {what:16s} {e0.elapsed_time(e1) / 3:.3f} ms")
        plan.check()


if __name__ == "__main__":
    main()
