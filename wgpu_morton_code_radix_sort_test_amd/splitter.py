"""Host-side mirror of `mrs_splitter_*`: where the global sorted order of a key array that is
spread over several ranks is cut.  No GPU is needed; the sharded sort on the GPUs
(`mrs_dist_sort_pairs`) runs the same C++ object on histograms from its own kernels.
"""
import ctypes

import numpy as np

from . import _lib

MAX_RANKS = 16


class SplitResultStruct(ctypes.Structure):
    _fields_ = [
        ("world", ctypes.c_int32),
        ("levels", ctypes.c_int32),
        ("total", ctypes.c_uint64),
        ("rank_size", ctypes.c_uint64 * 16),
        ("cut_key", ctypes.c_uint64 * 15),
        ("exact_mask", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("send_count", (ctypes.c_uint64 * 16) * 16),
    ]


class SplitResult:
    def __init__(self, raw):
        self.world = raw.world
        self.levels = raw.levels
        self.total = raw.total
        self.rank_size = [int(v) for v in raw.rank_size][: raw.world]
        self.cut_key = [int(v) for v in raw.cut_key][: raw.world - 1]
        self.exact_mask = int(raw.exact_mask)
        self.send_count = [[int(raw.send_count[s][d]) for d in range(raw.world)] for s in range(raw.world)]

    def key_class(self, keys):
        """2 * #{r : key >= cut_key[r]} - #{r : cut r is exact and key == cut_key[r]} (numpy array)."""
        k = np.asarray(keys).astype(np.uint64)
        c = np.zeros(k.shape, dtype=np.int64)
        for r, v in enumerate(self.cut_key):
            c += 2 * (k >= np.uint64(v))
            if (self.exact_mask >> r) & 1:
                c -= k == np.uint64(v)
        return c

    def recv_count(self, dst):
        return [self.send_count[s][dst] for s in range(self.world)]


class Splitter:
    """`mrs_splitter`: feed it the histograms it asks for until `request()` returns None."""

    def __init__(self, world, key_bits, tolerance):
        self.lib = _lib.load_library()
        self.world = world
        h = ctypes.c_void_p()
        _lib.check(self.lib.mrs_splitter_create(world, key_bits, tolerance, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self.lib.mrs_splitter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def request(self):
        """(shift, bits, [prefixes]) of the histogram wanted next, or None when the cuts are placed."""
        shift, bits, num = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        prefixes = (ctypes.c_uint64 * 15)()
        _lib.check(self.lib.mrs_splitter_next(self._h, ctypes.byref(shift), ctypes.byref(bits), ctypes.byref(num), prefixes))
        if bits.value == 0:
            return None
        return shift.value, bits.value, [int(prefixes[i]) for i in range(num.value)]

    def feed(self, counts):
        """counts: uint64 array [world][num_prefixes][1 << bits]."""
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        _lib.check(self.lib.mrs_splitter_feed(self._h, c.ctypes.data_as(ctypes.c_void_p)))

    def result(self):
        raw = SplitResultStruct()
        _lib.check(self.lib.mrs_splitter_get(self._h, ctypes.byref(raw)))
        return SplitResult(raw)


def local_histogram(keys, shift, bits, prefixes):
    """What one rank contributes to a request: uint64 [len(prefixes)][1 << bits]."""
    k = np.asarray(keys).astype(np.uint64)
    digit = ((k >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
    above = k >> np.uint64(shift + bits) if shift + bits < 64 else np.zeros_like(k)
    out = np.zeros((len(prefixes), 1 << bits), dtype=np.uint64)
    for j, p in enumerate(prefixes):
        sel = above == np.uint64(p)
        out[j] = np.bincount(digit[sel], minlength=1 << bits).astype(np.uint64)
    return out


def split_slices(slices, key_bits, tolerance):
    """Runs the splitter on key slices held in one process (tests, single-process tools)."""
    s = Splitter(len(slices), key_bits, tolerance)
    while True:
        req = s.request()
        if req is None:
            break
        shift, bits, prefixes = req
        s.feed(np.stack([local_histogram(k, shift, bits, prefixes) for k in slices]))
    res = s.result()
    s.close()
    return res
