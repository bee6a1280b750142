"""Host-side mirror of the C ABI: contexts and sort plans over torch device tensors.

torch is used for device memory and streams only; every computation is a call
into libmrs.so.  Buffer roles follow the reference's `Buffers` struct
(src/types.rs:68-77): `morton_codes`/`morton_codes_2` are keys A/B and
`indices`/`indices_2` are payload A/B of `SortPlan.sort_pairs`.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib

ALGO_THREE_KERNEL = 0
ALGO_ONESWEEP = 1
ALGO_HYBRID = 2

MORTON_UNIFORMS_BYTES = 51248
VERTEX_STRIDE = 48
TRIANGLE_STRIDE = 16


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("expected a device tensor")
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def _nbytes(t):
    return t.numel() * t.element_size()


def _stream(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream)


@dataclass
class PlanInfo:
    num_passes: int
    radix: int
    tile_items: int
    threads: int
    max_tiles: int
    histogram_len: int
    temp_bytes: int


class Context:
    """`mrs_context`: one per GPU (replaces device acquisition, src/main.rs:692-707)."""

    def __init__(self, device=0):
        self.lib = _lib.load_library()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        torch.cuda.set_device(self.device)
        handle = ctypes.c_void_p()
        _lib.check(self.lib.mrs_context_create(self.device.index, ctypes.byref(handle)))
        self._h = handle

    ATTR_COMPUTE_UNITS = 0
    ATTR_RANK_MODE = 1
    ATTR_DIE_AWARE_TICKETS = 2
    ATTR_NOALLOC_TIMEOUT = 3
    ATTR_ROCTX_RANGES = 4
    # aids for tests and A/B runs: a plan copies them from its context when it is created
    ATTR_TEST_HYBRID_MIN_N = 5
    ATTR_TEST_HYBRID_LEADING_BITS = 6
    ATTR_TEST_HYBRID_FORCE_FALLBACK = 7
    ATTR_TEST_NO_SMALL_SORT = 8
    ATTR_TEST_NO_PASS_SKIP = 9
    ATTR_TEST_WIDE_STATUS = 10
    ATTR_TEST_HYBRID_FORCE_ADAPTIVE = 11
    ATTR_TEST_HYBRID_NO_ADAPTIVE = 12
    ATTR_TEST_HYBRID_CAPACITY = 13
    ATTR_TEST_HYBRID_BIG_LOCAL = 14
    NOALLOC_MAX_N = 1 << 26

    def attribute(self, which):
        """`mrs_context_get_attribute`: what the context found on its device."""
        value = ctypes.c_int64()
        _lib.check(self.lib.mrs_context_get_attribute(self._h, int(which), ctypes.byref(value)))
        return int(value.value)

    def set_attribute(self, which, value):
        """`mrs_context_set_attribute`."""
        _lib.check(self.lib.mrs_context_set_attribute(self._h, int(which), int(value)))

    @property
    def rank_mode(self):
        """"atomic" (returning LDS atomics) or "match" (ballot matching)."""
        return "atomic" if self.attribute(self.ATTR_RANK_MODE) == 1 else "match"

    def close(self):
        if self._h:
            self.lib.mrs_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Morton encoders -------------------------------------------------
    def morton_encode30(self, xyz, mn, mx, keys_out=None, idx_out=None, stream=None, keys_only=False):
        """30-bit Morton keys + index payload of an [n, 3] f32 point tensor.  `keys_only`: no index
        array is written (for SortPlan.sort_pairs_indexed); the second value returned is None."""
        n = xyz.numel() // 3
        if xyz.dtype != torch.float32:
            raise ValueError("xyz must be float32")
        if keys_out is None:
            keys_out = torch.empty(n, dtype=torch.int32, device=self.device)
        if idx_out is None and not keys_only:
            idx_out = torch.empty(n, dtype=torch.int32, device=self.device)
        if keys_out.numel() < n or (idx_out is not None and idx_out.numel() < n):
            raise ValueError("output tensors are too small")
        c3 = ctypes.c_float * 3
        _lib.check(
            self.lib.mrs_morton_encode30(
                self._h, _ptr(xyz), n, c3(*[float(v) for v in mn]), c3(*[float(v) for v in mx]),
                _ptr(keys_out), _ptr(idx_out) if idx_out is not None else None, _stream(stream),
            )
        )
        return keys_out, idx_out

    def morton_encode_lut(self, uniforms, vertices, triangles, codes_out=None, stream=None):
        """Extended Morton codes from the reference's buffer layouts (bytes tensors)."""
        if _nbytes(uniforms) != MORTON_UNIFORMS_BYTES:
            raise ValueError(f"uniforms must be {MORTON_UNIFORMS_BYTES} bytes")
        if _nbytes(vertices) % VERTEX_STRIDE or _nbytes(triangles) % TRIANGLE_STRIDE:
            raise ValueError("vertices must have stride 48 and triangles stride 16")
        n = _nbytes(triangles) // TRIANGLE_STRIDE
        nv = _nbytes(vertices) // VERTEX_STRIDE
        if codes_out is None:
            codes_out = torch.empty(2 * n, dtype=torch.int32, device=self.device)
        if _nbytes(codes_out) < 8 * n:
            raise ValueError("codes_out is too small")
        _lib.check(
            self.lib.mrs_morton_encode_lut(
                self._h, _ptr(uniforms), _ptr(vertices), nv, _ptr(triangles), n, _ptr(codes_out), _stream(stream)
            )
        )
        return codes_out

    # ---- the sort without a plan and without allocations ---------------------------
    def sort_pairs_noalloc(self, keys_a, vals_a, keys_b, vals_b, n, begin_bit=0, end_bit=32, stream=None):
        """`mrs_sort_pairs_noalloc`: (u32 key, u32 payload) pairs, temporary memory in the library's
        static scratch array (n <= NOALLOC_MAX_N).  Returns True when the result is in the B tensors."""
        for t in (keys_a, vals_a, keys_b, vals_b):
            if t is None or _nbytes(t) < 4 * n:
                raise ValueError("tensor is missing or too small")
        in_b = ctypes.c_int(0)
        _lib.check(
            self.lib.mrs_sort_pairs_noalloc(self._h, n, _ptr(keys_a), _ptr(vals_a), _ptr(keys_b), _ptr(vals_b),
                                            begin_bit, end_bit, _stream(stream), ctypes.byref(in_b))
        )
        return bool(in_b.value)

    # ---- generators ----------------------------------------------------------
    def generate_points(self, seed, n, stream=None):
        xyz = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mrs_generate_points(self._h, seed, n, _ptr(xyz), _stream(stream)))
        return xyz

    def generate_u32(self, seed, stream_id, n, stream=None):
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.mrs_generate_u32(self._h, seed, stream_id, n, _ptr(out), _stream(stream)))
        return out

    # ---- multi-GPU building block ----------------------------------------------
    def bucket_histogram(self, keys, n, key_words, shift, bits, stream=None):
        counts = torch.empty(1 << bits, dtype=torch.int64, device=self.device)
        _lib.check(
            self.lib.mrs_bucket_histogram(self._h, _ptr(keys), n, key_words, shift, bits, _ptr(counts), _stream(stream))
        )
        return counts


class SortPlan:
    """`mrs_plan`: owns the temporaries of one sort shape (src/main.rs:31-85, 236-253)."""

    def __init__(self, ctx, n, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=0, end_bit=None,
                 algo=ALGO_ONESWEEP, tile_items=0):
        self.ctx = ctx
        self.lib = ctx.lib
        if end_bit is None:
            end_bit = 32 * key_words
        self.desc = _lib.PlanDesc(n, key_words, payload_bytes, digit_bits, begin_bit, end_bit, algo, tile_items)
        handle = ctypes.c_void_p()
        _lib.check(self.lib.mrs_plan_create(ctx._h, ctypes.byref(self.desc), ctypes.byref(handle)))
        self._h = handle
        raw = _lib.PlanInfoStruct()
        _lib.check(self.lib.mrs_plan_get_info(self._h, ctypes.byref(raw)))
        self.info = PlanInfo(raw.num_passes, raw.radix, raw.tile_items, raw.threads, raw.max_tiles,
                             raw.histogram_len, raw.temp_bytes)
        self.n = n
        self.key_words = key_words
        self.payload_bytes = payload_bytes

    def close(self):
        if self._h:
            self.lib.mrs_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_sizes(self, n, keys, vals):
        for k in keys:
            if k is not None and _nbytes(k) < 4 * self.key_words * n:
                raise ValueError("key tensor is too small")
        if self.payload_bytes:
            for v in vals:
                if v is None or _nbytes(v) < self.payload_bytes * n:
                    raise ValueError("payload tensor is missing or too small")

    def num_tiles(self, n):
        return (n + self.info.tile_items - 1) // self.info.tile_items

    def sort_pairs(self, keys_a, vals_a, keys_b, vals_b, n=None, stream=None):
        """Sorts; returns True when the result is in the B tensors (odd pass count)."""
        n = self.n if n is None else n
        self._check_sizes(n, (keys_a, keys_b), (vals_a, vals_b))
        in_b = ctypes.c_int(0)
        _lib.check(
            self.lib.mrs_sort_pairs(self._h, n, _ptr(keys_a), _ptr(vals_a), _ptr(keys_b), _ptr(vals_b),
                                    _stream(stream), ctypes.byref(in_b))
        )
        return bool(in_b.value)

    def sort_pairs_primed(self, keys_a, vals_a, keys_b, vals_b, n=None, stream=None):
        """`mrs_sort_pairs_primed`: sort_pairs that uses the histograms morton_encode30_for_sort left
        in this plan for exactly these keys (raises MrsError if the plan does not hold them)."""
        n = self.n if n is None else n
        self._check_sizes(n, (keys_a, keys_b), (vals_a, vals_b))
        in_b = ctypes.c_int(0)
        _lib.check(
            self.lib.mrs_sort_pairs_primed(self._h, n, _ptr(keys_a), _ptr(vals_a), _ptr(keys_b), _ptr(vals_b),
                                           _stream(stream), ctypes.byref(in_b))
        )
        return bool(in_b.value)

    def sort_pairs_indexed(self, keys_a, vals_a, keys_b, vals_b, n=None, stream=None):
        """`mrs_sort_pairs_indexed`: sort_pairs whose input payload is the element index; vals_a is
        ping-pong space and need not hold anything.  Returns True when the result is in the B tensors."""
        n = self.n if n is None else n
        self._check_sizes(n, (keys_a, keys_b), (vals_a, vals_b))
        in_b = ctypes.c_int(0)
        _lib.check(
            self.lib.mrs_sort_pairs_indexed(self._h, n, _ptr(keys_a), _ptr(vals_a), _ptr(keys_b), _ptr(vals_b),
                                            _stream(stream), ctypes.byref(in_b))
        )
        return bool(in_b.value)

    def morton_encode30_for_sort(self, xyz, mn, mx, keys_out=None, idx_out=None, stream=None):
        """`mrs_morton_encode30_for_sort`: the encoder that also leaves the digit histograms of the
        keys in this plan; pass the returned key tensor to sort_pairs_primed as keys_a, unchanged."""
        n = xyz.numel() // 3
        if xyz.dtype != torch.float32:
            raise ValueError("xyz must be float32")
        if keys_out is None:
            keys_out = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        if idx_out is None:
            idx_out = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        if keys_out.numel() < n or idx_out.numel() < n:
            raise ValueError("output tensors are too small")
        c3 = ctypes.c_float * 3
        _lib.check(
            self.lib.mrs_morton_encode30_for_sort(
                self._h, _ptr(xyz), n, c3(*[float(v) for v in mn]), c3(*[float(v) for v in mx]),
                _ptr(keys_out), _ptr(idx_out), _stream(stream),
            )
        )
        return keys_out, idx_out

    def sort_gather(self, keys_a, keys_b, records_in, records_out, record_bytes, n=None, index_a=None, index_b=None,
                    stream=None):
        """`mrs_sort_gather`: sorts (key, source index) pairs and moves each record of
        `record_bytes` once.  Returns (result_in_b, sorted source indices)."""
        n = self.n if n is None else n
        if self.payload_bytes != 4:
            raise ValueError("sort_gather needs a plan with payload_bytes=4")
        if index_a is None:
            index_a = torch.empty(max(1, n), dtype=torch.int32, device=self.ctx.device)
        if index_b is None:
            index_b = torch.empty(max(1, n), dtype=torch.int32, device=self.ctx.device)
        self._check_sizes(n, (keys_a, keys_b), (index_a, index_b))
        if _nbytes(records_in) < n * record_bytes or _nbytes(records_out) < n * record_bytes:
            raise ValueError("record buffers are too small")
        in_b = ctypes.c_int(0)
        _lib.check(
            self.lib.mrs_sort_gather(self._h, n, _ptr(keys_a), _ptr(keys_b), _ptr(index_a), _ptr(index_b),
                                     _ptr(records_in), _ptr(records_out), record_bytes, _stream(stream),
                                     ctypes.byref(in_b))
        )
        return bool(in_b.value), (index_b if in_b.value else index_a)

    def histogram(self, keys, pass_number, n=None, hist_out=None, stream=None):
        n = self.n if n is None else n
        self._check_sizes(n, (keys,), ())
        length = self.info.radix * self.num_tiles(n)
        if hist_out is None:
            hist_out = torch.empty(length, dtype=torch.int32, device=self.ctx.device)
        if hist_out.numel() < length:
            raise ValueError("hist_out is too small")
        _lib.check(self.lib.mrs_histogram(self._h, n, _ptr(keys), pass_number, _ptr(hist_out), _stream(stream)))
        return hist_out

    def scan(self, hist, n=None, stream=None):
        n = self.n if n is None else n
        if hist.numel() < self.info.radix * self.num_tiles(n):
            raise ValueError("hist is too small")
        _lib.check(self.lib.mrs_scan(self._h, n, _ptr(hist), _stream(stream)))
        return hist

    def scatter(self, keys_in, vals_in, keys_out, vals_out, scanned, pass_number, n=None, stream=None):
        n = self.n if n is None else n
        self._check_sizes(n, (keys_in, keys_out), (vals_in, vals_out))
        if scanned.numel() < self.info.radix * self.num_tiles(n):
            raise ValueError("scanned histogram is too small")
        _lib.check(
            self.lib.mrs_scatter(self._h, n, _ptr(keys_in), _ptr(vals_in), _ptr(keys_out), _ptr(vals_out),
                                 _ptr(scanned), pass_number, _stream(stream))
        )

    def partition(self, keys_in, vals_in, keys_out, vals_out, shift, bits, bucket_map, num_parts, n=None,
                  stream=None):
        """Stable partition by bucket_map[(key >> shift) & mask]; returns u32[16] part sizes (device)."""
        n = self.n if n is None else n
        self._check_sizes(n, (keys_in, keys_out), (vals_in, vals_out))
        if bucket_map.dtype != torch.uint8 or bucket_map.numel() < (1 << bits):
            raise ValueError("bucket_map must be a uint8 tensor with 1 << bits entries")
        counts = torch.empty(16, dtype=torch.int32, device=self.ctx.device)
        _lib.check(
            self.lib.mrs_partition(self._h, n, _ptr(keys_in), _ptr(vals_in), _ptr(keys_out), _ptr(vals_out),
                                   shift, bits, _ptr(bucket_map), num_parts, _ptr(counts), _stream(stream))
        )
        return counts

    def set_timing(self, enable=True):
        _lib.check(self.lib.mrs_plan_set_timing(self._h, 1 if enable else 0))

    def timings(self):
        """[(label, milliseconds)] for every launch of the last sort_pairs call (timing must be on)."""
        cap = 96
        ms = (ctypes.c_float * cap)()
        labels = ctypes.create_string_buffer(32 * cap)
        count = ctypes.c_int(0)
        _lib.check(self.lib.mrs_plan_get_timings(self._h, cap, ms, labels, ctypes.byref(count)))
        out = []
        for i in range(count.value):
            raw = labels.raw[32 * i: 32 * i + 32]
            out.append((raw.split(b"\0", 1)[0].decode(), float(ms[i])))
        return out

    def check(self, stream=None):
        """Synchronises and raises MrsError if a look-back timed out on the device."""
        _lib.check(self.lib.mrs_plan_check(self._h, _stream(stream)))

    def passes_left_out(self, stream=None):
        """Synchronises; bit p is set if pass p of the last sort_pairs did not run on the device
        (64-bit keys whose digit p is the same for every key)."""
        mask = ctypes.c_uint32(0)
        _lib.check(self.lib.mrs_plan_get_passes_left_out(self._h, _stream(stream), ctypes.byref(mask)))
        return mask.value

    SCHEDULE_PASSES = 0
    SCHEDULE_MSD = 1
    SCHEDULE_MSD_REFUSED = 2
    SCHEDULE_MSD_ADAPTIVE = 3

    def schedule(self, stream=None):
        """`mrs_plan_get_schedule`: synchronises; (schedule the last sort ran, largest sub-bucket the
        device counted).  Only ALGO_HYBRID plans ever report anything but SCHEDULE_PASSES."""
        which, largest = ctypes.c_int32(0), ctypes.c_uint32(0)
        _lib.check(self.lib.mrs_plan_get_schedule(self._h, _stream(stream), ctypes.byref(which), ctypes.byref(largest)))
        return which.value, largest.value

    def oversized(self, stream=None):
        """`mrs_plan_get_oversized`: synchronises; (sub-buckets the last three-move sort found too large
        for a workgroup and sorted with its two extra passes, pairs they held, pairs a workgroup held)."""
        subs, pairs, cap = ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
        _lib.check(self.lib.mrs_plan_get_oversized(self._h, _stream(stream), ctypes.byref(subs), ctypes.byref(pairs),
                                                   ctypes.byref(cap)))
        return subs.value, pairs.value, cap.value

    def adaptive(self, stream=None):
        """`mrs_plan_get_adaptive`: synchronises; (buckets, groups) the last sort was cut into if it
        ran SCHEDULE_MSD_ADAPTIVE, else zeros."""
        buckets, groups = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _lib.check(self.lib.mrs_plan_get_adaptive(self._h, _stream(stream), ctypes.byref(buckets), ctypes.byref(groups)))
        return buckets.value, groups.value
