#!/usr/bin/env python3
"""Phase breakdown of the onesweep digit pass from a diagnostic build (-DMRS_STAMPS).

    MRS_LIBRARY_PATH=scratch/libmrs_diag.so python3 tools/phase_stamps.py --log2n 28

Stamps (100 MHz clock, thread 0 of every tile of pass 1):
 0 kernel entry  1 ticket known  2 keys loaded  3 ranked  4 digit offsets ready
 5 look-back done, keys staged in LDS, payload requested  6 workgroup barrier passed
 7 all stores retired
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402

NAMES = ["-", "key wait", "rank", "hist+bases", "lookback+stage", "barrier wait", "write_out"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--tile-items", type=int, default=0)
    args = ap.parse_args()
    ctx = mrs.Context(0)
    lib = ctx.lib
    n = 1 << args.log2n
    plan = mrs.SortPlan(ctx, n, tile_items=args.tile_items)
    tiles = plan.info.max_tiles
    stamps = torch.zeros(tiles * 16, dtype=torch.int64, device=ctx.device)
    lib.mrs_debug_set_stamps.argtypes = [ctypes.c_void_p]
    lib.mrs_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    kb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
    for it in range(2):
        xyz = ctx.generate_points(7 + it, n)
        keys, idx = ctx.morton_encode30(xyz, [0, 0, 0], [1, 1, 1])
        del xyz
        stamps.zero_()
        plan.sort_pairs(keys, idx, kb, vb)
        plan.check()
    raw = stamps.cpu().numpy().reshape(tiles, 16)
    st = raw[:, :8].astype(np.float64) / 100.0  # microseconds
    t0 = st[:, 0].min()
    print("tiles", tiles, "pass span us", st[:, 7].max() - t0)
    d = np.diff(st, axis=1)
    for k, name in enumerate(NAMES):
        col = d[:, k]
        print(f"{name:12s} mean {col.mean():8.2f}  p10 {np.percentile(col, 10):8.2f}  p50 {np.percentile(col, 50):8.2f}  "
              f"p90 {np.percentile(col, 90):8.2f}  max {col.max():8.2f}")
    # finer stamps: 11 first look-back round reduced, 12 look-back finished (thread 0's wave),
    # 13 key stores issued, 14 payload staged in LDS
    fine = raw[:, 11:15].astype(np.float64) / 100.0
    have = raw[:, 11] != 0
    if have.any():
        for name, col in (("  lb round 1", fine[:, 0] - st[:, 4]), ("  lb more rounds", fine[:, 1] - fine[:, 0]),
                          ("  stage+val issue", st[:, 5] - fine[:, 1]), ("  barrier wait", st[:, 6] - st[:, 5]),
                          ("  key stores", fine[:, 2] - st[:, 6]), ("  val staged", fine[:, 3] - fine[:, 2]),
                          ("  val stores", st[:, 7] - fine[:, 3])):
            col = col[have]
            print(f"{name:18s} mean {col.mean():8.2f}  p10 {np.percentile(col, 10):8.2f}  p50 {np.percentile(col, 50):8.2f}  "
                  f"p90 {np.percentile(col, 90):8.2f}")
    life = st[:, 7] - st[:, 0]
    print(f"{'tile life':12s} mean {life.mean():8.2f}  p50 {np.percentile(life, 50):8.2f}  p90 {np.percentile(life, 90):8.2f}")
    for k, name in ((8, "lookback round trips"), (9, "tiles walked"), (10, "empty rounds")):
        col = raw[:, k].astype(np.float64)
        print(f"{name:22s} mean {col.mean():7.2f}  p50 {np.percentile(col, 50):6.1f}  p90 {np.percentile(col, 90):6.1f}  max {col.max():6.0f}")
    # concurrency: average number of tiles alive
    print("avg tiles in flight", life.sum() / (st[:, 7].max() - t0))
    # start order vs tile id
    order = np.argsort(st[:, 0])
    print("max |start rank - tile id|", np.abs(order - np.arange(tiles)).max())


if __name__ == "__main__":
    main()
