"""Python mirror of the reference's entry points for the compute path.

`run_compute_shaders` (src/main.rs:376-688) builds the scene, allocates the
`Buffers` set and records "Morton Coding" followed by NUM_PASSES radix passes;
`run_headless` (src/main.rs:690-1033) submits, reads both key buffers back, picks
the output by pass parity and checks adjacent pairs.  Here the recording is a
sequence of asynchronous libmrs calls on one HIP stream.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import scene
from .api import ALGO_ONESWEEP, Context, SortPlan


@dataclass
class Buffers:
    """Device buffer set of src/types.rs:68-77 (prefix_sums live inside the plan)."""
    vertices: torch.Tensor
    indices: torch.Tensor
    indices_2: torch.Tensor
    morton_uniforms: torch.Tensor
    morton_codes: torch.Tensor
    morton_codes_2: torch.Tensor
    plan: SortPlan
    result_in_2: bool = False
    triangles_in_2: bool = False  # where the sorted TriangleIndices are (follows result_in_2 unless gathered)
    source_index: torch.Tensor = None  # gather mode: sorted source indices (u32)
    host: dict = field(default_factory=dict)


def create_all_buffers(ctx, positions, indices, generator, num_passes, algo, tile_items=0, gather_payload=False):
    """create_all_buffers (src/main.rs:31-85): uploads vertices/indices/uniforms and
    allocates the second index buffer and the two Morton code buffers."""
    dev = ctx.device
    n = indices.shape[0]
    blob = scene.create_morton_uniforms(generator)
    vertices_b = torch.from_numpy(scene.pack_vertices(positions)).to(dev)
    indices_b = torch.from_numpy(scene.pack_triangles(indices)).to(dev)
    plan = SortPlan(ctx, n, key_words=2, payload_bytes=4 if gather_payload else 16, digit_bits=scene.BITS_PER_PASS,
                    begin_bit=0, end_bit=8 * num_passes, algo=algo, tile_items=tile_items)
    return Buffers(
        vertices=vertices_b,
        indices=indices_b,
        indices_2=torch.empty_like(indices_b),
        morton_uniforms=torch.from_numpy(blob).to(dev),
        morton_codes=torch.empty(2 * n, dtype=torch.int32, device=dev),
        morton_codes_2=torch.empty(2 * n, dtype=torch.int32, device=dev),
        plan=plan,
        host={"positions": positions, "indices": indices, "uniforms": blob},
    )


def run_compute_shaders(ctx, stream=None, num_vertices=scene.NUM_VERTICES, num_triangles=scene.NUM_TRIANGLES,
                        num_passes=scene.NUM_PASSES, seed=scene.RNG_SEED, algo=ALGO_ONESWEEP, tile_items=0,
                        gather_payload=False, scene_rng="counter"):
    """Enqueues Morton coding + the radix sort on `stream`; returns the Buffers.

    Like the reference function it only records work: nothing is synchronised.  With
    `gather_payload` the TriangleIndices records are not carried through the passes: (code,
    source index) pairs are sorted and every record moves once, into `indices_2`.
    """
    # scene_rng="chacha8": the reference's ChaCha8Rng stream as restated in scene.ChaCha8Rng
    # (unverified against the Rust crates); "counter": this package's counter-based generator
    make_scene = scene.create_scene_chacha8 if scene_rng == "chacha8" else scene.create_scene
    positions, indices, generator = make_scene(num_vertices, num_triangles, seed)
    buffers = create_all_buffers(ctx, positions, indices, generator, num_passes, algo, tile_items, gather_payload)
    # "Morton Coding", src/main.rs:581-592
    ctx.morton_encode_lut(buffers.morton_uniforms, buffers.vertices, buffers.indices, buffers.morton_codes,
                          stream=stream)
    # NUM_PASSES x {histogram, prefix sums, indexing}, src/main.rs:594-665
    if gather_payload:
        buffers.result_in_2, buffers.source_index = buffers.plan.sort_gather(
            buffers.morton_codes, buffers.morton_codes_2, buffers.indices, buffers.indices_2, 16, stream=stream)
        buffers.triangles_in_2 = True
    else:
        buffers.result_in_2 = buffers.plan.sort_pairs(buffers.morton_codes, buffers.indices, buffers.morton_codes_2,
                                                      buffers.indices_2, stream=stream)
        buffers.triangles_in_2 = buffers.result_in_2
    return buffers


def select_bits(value, i):
    """src/main.rs:1042-1051"""
    if i < 1 or i > 8:
        raise ValueError("Input must be between 1 and 8")
    return value if i == 8 else value & ((np.uint64(1) << np.uint64(i * 8)) - np.uint64(1))


def run_headless(device=0, **kwargs):
    """Runs the path once and applies the reference's readback check (src/main.rs:819-864).

    Returns a dict: sorted, times_unsorted, all_zero, keys (u64 array), triangles ([n,3] u32).
    """
    ctx = Context(device)
    num_passes = kwargs.get("num_passes", scene.NUM_PASSES)
    buffers = run_compute_shaders(ctx, **kwargs)
    buffers.plan.check()  # synchronises; raises on a device-side timeout
    out_codes = buffers.morton_codes_2 if buffers.result_in_2 else buffers.morton_codes
    out_tris = buffers.indices_2 if buffers.triangles_in_2 else buffers.indices
    keys = out_codes.cpu().numpy().view(np.uint64)
    tris = out_tris.cpu().numpy().view(np.uint32).reshape(-1, 4)[:, :3]
    sel = select_bits(keys, num_passes)
    times_unsorted = int(np.count_nonzero(sel[:-1] > sel[1:]))
    return {
        "sorted": times_unsorted == 0,
        "times_unsorted": times_unsorted,
        "all_zero": bool(np.all(sel == 0)),
        "keys": keys,
        "triangles": tris,
        "buffers": buffers,
    }
