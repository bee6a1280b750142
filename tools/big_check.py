#!/usr/bin/env python3
"""Sorts of more than 2^31 pairs (the library takes up to 2^32 - 2), checked through size-independent
properties in slices of 2^28 elements: ascending keys, ties in payload order (payload = source index),
every payload pointing at its key, and the pair checksum of the input kept."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402
from wgpu_morton_code_radix_sort_test_amd.dist import pair_checksum  # noqa: E402

STEP = 1 << 28


def u(x):
    return x.to(torch.int64) & 0xFFFFFFFF


def check(keys_in, ok, ov, n):
    good, total_in, total_out = True, 0, 0
    prev = None
    for at in range(0, n, STEP):
        end = min(n, at + STEP)
        k, v = u(ok[at:end]), u(ov[at:end])
        dk = k[1:] - k[:-1]
        good &= bool(((dk > 0) | ((dk == 0) & (v[1:] > v[:-1]))).all())
        if prev is not None:
            good &= prev < (int(k[0]), int(v[0]))
        prev = (int(k[-1]), int(v[-1]))
        good &= bool((u(keys_in[v]) == k).all())
        total_out = (total_out + pair_checksum(ok[at:end], ov[at:end])) & ((1 << 64) - 1)
        idx = torch.arange(at, end, dtype=torch.int64, device=ok.device).to(torch.int32)
        total_in = (total_in + pair_checksum(keys_in[at:end], idx)) & ((1 << 64) - 1)
        del k, v, dk, idx
    return good and total_in == total_out


def check64(words_in, ok, ov, n):
    """The same for 64-bit keys (two words each, the high word second)."""
    good, prev, index_sum = True, None, 0
    for at in range(0, n, STEP):
        end = min(n, at + STEP)
        w = ok[2 * at:2 * end].view(-1, 2)
        kl, kh, v = u(w[:, 0]), u(w[:, 1]), u(ov[at:end])
        up = kh[1:] > kh[:-1]
        same_h = kh[1:] == kh[:-1]
        good &= bool((up | (same_h & (kl[1:] > kl[:-1])) | (same_h & (kl[1:] == kl[:-1]) & (v[1:] > v[:-1]))).all())
        first = (int(kh[0]), int(kl[0]), int(v[0]))
        if prev is not None:
            good &= prev < first
        prev = (int(kh[-1]), int(kl[-1]), int(v[-1]))
        src = words_in.view(-1, 2)[v]
        good &= bool((u(src[:, 0]) == kl).all() and (u(src[:, 1]) == kh).all())
        index_sum += int(v.sum())
        del w, kl, kh, v, up, same_h, src
    return good and index_sum == n * (n - 1) // 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-words", type=int, default=1)
    ap.add_argument("--n", type=int, nargs="+", default=[3 << 30, (1 << 32) - 2])
    ap.add_argument("--algo", default="onesweep")
    ap.add_argument("--mask", type=lambda x: int(x, 0), default=0xFFFFFFFF, help="AND mask on the random keys")
    args = ap.parse_args()
    ctx = mrs.Context(0)
    algo = {"onesweep": mrs.ALGO_ONESWEEP, "three_kernel": mrs.ALGO_THREE_KERNEL}[args.algo]
    bad = False
    for n in args.n:
        keys = ctx.generate_u32(9, 1, n * args.key_words)
        if args.mask != 0xFFFFFFFF:
            keys &= args.mask - (1 << 32) if args.mask >= (1 << 31) else args.mask
        ka, va = keys.clone(), torch.empty(n, dtype=torch.int32, device=ctx.device)
        kb, vb = torch.empty_like(ka), torch.empty_like(va)
        plan = mrs.SortPlan(ctx, n, key_words=args.key_words, payload_bytes=4, digit_bits=8, algo=algo)
        in_b = plan.sort_pairs_indexed(ka, va, kb, vb)
        plan.check()
        torch.cuda.synchronize()
        rk, rv = (kb, vb) if in_b else (ka, va)
        good = check(keys, rk, rv, n) if args.key_words == 1 else check64(keys, rk, rv, n)
        print(f"n = {n}: {'ok' if good else 'MISMATCH'}", flush=True)
        bad |= not good
        plan.close()
        del keys, ka, va, kb, vb
        torch.cuda.empty_cache()
    print("BIG CHECK", "FAILED" if bad else "ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
