#!/usr/bin/env python3
"""Times the LUT Morton encoder (the reference's "Morton Coding" dispatch) at a large size."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-triangles", type=int, default=26)
    ap.add_argument("--log2-vertices", type=int, default=24)
    args = ap.parse_args()
    ctx = mrs.Context(0)
    nv, nt = 1 << args.log2_vertices, 1 << args.log2_triangles
    gen = mrs.scene.StandInMortonGenerator([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    blob = torch.from_numpy(mrs.scene.create_morton_uniforms(gen)).to(ctx.device)
    g = torch.Generator(device=ctx.device).manual_seed(1)
    verts = torch.rand((nv, 12), generator=g, device=ctx.device, dtype=torch.float32)  # stride-48 vertices
    for locality, name in ((False, "random vertex indices"), (True, "vertex indices near the triangle index")):
        if locality:
            base = (torch.arange(nt, device=ctx.device, dtype=torch.int64) * nv // nt).unsqueeze(1)
            idx = (base + torch.randint(0, 64, (nt, 3), generator=g, device=ctx.device)).clamp_(max=nv - 1)
        else:
            idx = torch.randint(0, nv, (nt, 3), generator=g, device=ctx.device)
        tris = torch.zeros((nt, 4), dtype=torch.int32, device=ctx.device)
        tris[:, :3] = idx.to(torch.int32)
        codes = ctx.morton_encode_lut(blob, verts.view(torch.int32).reshape(-1), tris.reshape(-1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.morton_encode_lut(blob, verts.view(torch.int32).reshape(-1), tris.reshape(-1), codes)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        algo = nt * (16 + 36 + 8)
        print(f"{name}: {nt} triangles, {nv} vertices: {ms:.3f} ms, {nt / ms / 1e6:.2f} Gtriangles/s, "
              f"{algo / ms / 1e6:.0f} GB/s algorithmic (16 B triangle + 3 x 12 B positions + 8 B code)")


if __name__ == "__main__":
    main()
