"""ctypes wrapper of the CPU oracle (oracle/libmrs_oracle.so) for the tests.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg load it.  The library is built from oracle/ on first use when
it is missing (gcc only; no GPU needed).
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libmrs_oracle.so")
CPUSORT = os.path.join(ORACLE_DIR, "libmrs_cpusort.so")

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_int = ctypes.c_int


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def _void_or(arr):
    return arr.ctypes.data_as(ctypes.c_void_p) if arr is not None else None


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        self.L = L
        L.orc_div_ceil.restype = _u64
        L.orc_div_ceil.argtypes = [_u64, _u64]
        L.orc_num_workgroups.restype = _u64
        L.orc_num_workgroups.argtypes = [_u64, _u64]
        L.orc_round_up.restype = _u64
        L.orc_round_up.argtypes = [_u64, _u64]
        L.orc_prefix_buffer_sizes.restype = _int
        L.orc_prefix_buffer_sizes.argtypes = [_u64, _u64p, _int]
        L.orc_select_digit8.restype = _u32
        L.orc_select_digit8.argtypes = [_u32, _u32, _u32]
        L.orc_select_digit6.restype = _u32
        L.orc_select_digit6.argtypes = [_u32, _u32, _u32]
        L.orc_select_bits.restype = _u64
        L.orc_select_bits.argtypes = [_u64, _u32]
        L.orc_block_histogram.restype = None
        L.orc_block_histogram.argtypes = [_u32p, _u64, _int, _u32, _u32, _u32, _u32p]
        L.orc_exclusive_scan.restype = None
        L.orc_exclusive_scan.argtypes = [_u32p, _u64]
        L.orc_hierarchical_scan.restype = None
        L.orc_hierarchical_scan.argtypes = [_u32p, _u64]
        L.orc_ballot_rank_256.restype = None
        L.orc_ballot_rank_256.argtypes = [_u32p, _u32, _u32p]
        L.orc_scatter.restype = None
        L.orc_scatter.argtypes = [_u32p, ctypes.c_void_p, _u32p, ctypes.c_void_p, _u64, _int, _int, _u32, _u32, _u32, _u32p]
        L.orc_lsd_sort.restype = _int
        L.orc_lsd_sort.argtypes = [_u32p, ctypes.c_void_p, _u32p, ctypes.c_void_p, _u64, _int, _int, _u32, _u32, _u32, _u32, _u32p]
        L.orc_stable_sort_pairs.restype = None
        L.orc_stable_sort_pairs.argtypes = [_u32p, ctypes.c_void_p, _u32p, ctypes.c_void_p, _u64, _int, _int, _u32, _u32]
        L.orc_count_unsorted.restype = _u64
        L.orc_count_unsorted.argtypes = [_u32p, _u64, _int, _u32]
        L.orc_morton_encode_lut.restype = None
        L.orc_morton_encode_lut.argtypes = [_u8p, _u8p, _u64, _u8p, _u64, _u32p]
        L.orc_morton_encode30.restype = None
        L.orc_morton_encode30.argtypes = [_f32p, _u64, _f32p, _f32p, _u32p, _u32p]
        L.orc_rand_u32.restype = _u32
        L.orc_rand_u32.argtypes = [_u64, _u64, _u64]
        L.orc_gen_points.restype = None
        L.orc_gen_points.argtypes = [_u64, _u64, _f32p]
        L.orc_gen_u32.restype = None
        L.orc_gen_u32.argtypes = [_u64, _u64, _u64, _u32p]
        L.orc_minmax3.restype = None
        L.orc_minmax3.argtypes = [_f32p, _u64, _f32p, _f32p]
        for name in ("orc_selftest_workgroup_calculation", "orc_selftest_digit6", "orc_selftest_digit8"):
            getattr(L, name).restype = _int
            getattr(L, name).argtypes = []
        L.orc_selftest_workgroup_local_index.restype = _int
        L.orc_selftest_workgroup_local_index.argtypes = [_u32]

    # ---- convenience wrappers over numpy arrays -------------------------------
    def prefix_buffer_sizes(self, n):
        out = np.zeros(8, dtype=np.uint64)
        k = self.L.orc_prefix_buffer_sizes(n, out, 8)
        return [int(v) for v in out[:k]]

    def gen_points(self, seed, n):
        xyz = np.empty((n, 3), dtype=np.float32)
        self.L.orc_gen_points(seed, n, xyz.reshape(-1))
        return xyz

    def gen_u32(self, seed, stream, n):
        out = np.empty(n, dtype=np.uint32)
        self.L.orc_gen_u32(seed, stream, n, out)
        return out

    def minmax3(self, xyz):
        mn = np.empty(3, dtype=np.float32)
        mx = np.empty(3, dtype=np.float32)
        self.L.orc_minmax3(np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1), xyz.shape[0], mn, mx)
        return mn, mx

    def morton_encode30(self, xyz, mn, mx):
        n = xyz.shape[0]
        keys = np.empty(n, dtype=np.uint32)
        idx = np.empty(n, dtype=np.uint32)
        self.L.orc_morton_encode30(np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1), n,
                                   np.ascontiguousarray(mn, dtype=np.float32), np.ascontiguousarray(mx, dtype=np.float32),
                                   keys, idx)
        return keys, idx

    def morton_encode_lut(self, blob, vertices, triangles):
        n = triangles.size // 16
        codes = np.empty(2 * n, dtype=np.uint32)
        self.L.orc_morton_encode_lut(blob, vertices, vertices.size // 48, triangles, n, codes)
        return codes

    def block_histogram(self, keys, n, key_words, shift, bits, tile):
        W = (n + tile - 1) // tile
        hist = np.zeros(max(1, (1 << bits) * W), dtype=np.uint32)
        self.L.orc_block_histogram(keys, n, key_words, shift, bits, tile, hist)
        return hist

    def exclusive_scan(self, vals):
        out = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self.L.orc_exclusive_scan(out, out.size)
        return out

    def hierarchical_scan(self, vals):
        out = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self.L.orc_hierarchical_scan(out, out.size)
        return out

    def ballot_rank_256(self, digits, nvalid=256):
        d = np.zeros(256, dtype=np.uint32)
        d[: len(digits)] = digits
        out = np.zeros(256, dtype=np.uint32)
        self.L.orc_ballot_rank_256(d, nvalid, out)
        return out

    def scatter(self, keys, vals, n, key_words, payload_words, shift, bits, tile, scanned):
        kout = np.zeros_like(keys)
        vout = np.zeros_like(vals) if vals is not None else None
        self.L.orc_scatter(keys, _void_or(vals), kout, _void_or(vout), n, key_words, payload_words, shift, bits, tile,
                           scanned)
        return kout, vout

    def lsd_sort(self, keys, vals, n, key_words, payload_words, bits, begin_bit, end_bit, tile):
        """Returns (sorted keys, sorted vals, result_in_b)."""
        ka = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        kb = np.zeros_like(ka)
        va = np.ascontiguousarray(vals, dtype=np.uint32).copy() if vals is not None else None
        vb = np.zeros_like(va) if va is not None else None
        W = (n + tile - 1) // tile
        hist = np.zeros(max(1, (1 << bits) * W), dtype=np.uint32)
        in_b = self.L.orc_lsd_sort(ka, _void_or(va), kb, _void_or(vb), n, key_words, payload_words, bits, begin_bit,
                                   end_bit, tile, hist)
        return (kb, vb, True) if in_b else (ka, va, False)

    def stable_sort_pairs(self, keys, vals, n, key_words, payload_words, begin_bit, end_bit):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        kout = np.zeros_like(keys)
        vout = np.zeros_like(vals) if vals is not None else None
        self.L.orc_stable_sort_pairs(keys, _void_or(vals), kout, _void_or(vout), n, key_words, payload_words, begin_bit,
                                     end_bit)
        return kout, vout

    def count_unsorted(self, keys, n, key_words, num_passes_of_8):
        return int(self.L.orc_count_unsorted(np.ascontiguousarray(keys, dtype=np.uint32), n, key_words, num_passes_of_8))


class CpuSort:
    """std::sort / __gnu_parallel::sort baseline (oracle/cpu_sort.cpp)."""

    def __init__(self):
        if not os.path.exists(CPUSORT):
            build()
        L = ctypes.CDLL(CPUSORT)
        L.cpu_sort_pairs_u32.restype = ctypes.c_double
        L.cpu_sort_pairs_u32.argtypes = [_u32p, _u32p, _u64, _int, ctypes.POINTER(_int)]
        L.cpu_hardware_threads.restype = _int
        L.cpu_hardware_threads.argtypes = []
        self.L = L

    def sort_pairs(self, keys, idx, threads):
        used = _int(0)
        secs = self.L.cpu_sort_pairs_u32(keys, idx, keys.size, threads, ctypes.byref(used))
        return secs, used.value

    def hardware_threads(self):
        return self.L.cpu_hardware_threads()
