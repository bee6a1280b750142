"""Seeded test inputs shared by the CPU and GPU tests."""
import numpy as np

DISTRIBUTIONS = ("random", "all_equal", "sorted", "reversed", "one_hot_digit", "few_distinct", "morton30")


def make_keys(oracle, n, key_words, dist, seed=7):
    """u32 array of n*key_words words ((lo, hi) interleaved for 64-bit keys)."""
    words = n * key_words
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    if dist == "random":
        k = oracle.gen_u32(seed, 31, words)
    elif dist == "all_equal":
        k = np.full(words, 0xDEADBEEF, dtype=np.uint32)
    elif dist in ("sorted", "reversed"):
        if key_words == 1:
            k = np.sort(oracle.gen_u32(seed, 32, n))
        else:
            v = np.sort(oracle.gen_u32(seed, 32, words).view(np.uint64))
            k = v.view(np.uint32)
        if dist == "reversed":
            k = (k.view(np.uint64)[::-1].copy().view(np.uint32) if key_words == 2 else k[::-1].copy())
    elif dist == "one_hot_digit":
        # only one 8-bit digit varies (bits 8..15); every other digit is constant
        r = oracle.gen_u32(seed, 33, n)
        lo = (np.uint32(0x12000034) | ((r & np.uint32(0xFF)) << np.uint32(8))).astype(np.uint32)
        if key_words == 1:
            k = lo
        else:
            k = np.empty(words, dtype=np.uint32)
            k[0::2] = lo
            k[1::2] = np.uint32(0x00C0FFEE)
    elif dist == "few_distinct":
        r = oracle.gen_u32(seed, 34, words)
        k = (r % np.uint32(5)) * np.uint32(0x01010101)
    elif dist == "morton30":
        xyz = oracle.gen_points(seed, n)
        mn, mx = oracle.minmax3(xyz)
        k30, _ = oracle.morton_encode30(xyz, mn, mx)
        if key_words == 1:
            k = k30
        else:
            k = np.zeros(words, dtype=np.uint32)
            k[0::2] = k30
            k[1::2] = k30 >> np.uint32(7)
    else:
        raise ValueError(dist)
    return np.ascontiguousarray(k, dtype=np.uint32)


def make_vals(oracle, n, payload_words, seed=7):
    """Payload: word 0 = original index; further words (16-byte payload) random."""
    if payload_words == 0:
        return None
    v = np.zeros((n, payload_words), dtype=np.uint32)
    v[:, 0] = np.arange(n, dtype=np.uint32)
    for w in range(1, payload_words):
        v[:, w] = oracle.gen_u32(seed, 40 + w, n) if n else 0
    return np.ascontiguousarray(v.reshape(-1))
