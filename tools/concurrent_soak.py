#!/usr/bin/env python3
"""Host threads that sort on one GPU at the same time, each with a context, a stream and a plan of
its own, started together round after round so that their calls begin within microseconds of each
other.  Every sort is followed by the plan's error check (a look-back that starved reports a
time-out there) and by a check of the output on the device.  Meant for the rule that at most one
sort per device deals its tiles per die at a time: two of them can hold the compute units the
other needs."""
import argparse
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--log2n", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--sorts-per-round", type=int, default=4)
    args = ap.parse_args()
    n = 1 << args.log2n
    gate = threading.Barrier(args.threads)
    errors, bad = [], [0] * args.threads

    def body(t):
        try:
            torch.cuda.set_device(0)
            ctx = mrs.Context(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                plan = mrs.SortPlan(ctx, n)
                kb = torch.empty(n, dtype=torch.int32, device=ctx.device)
                vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
                for rnd in range(args.rounds):
                    keys = ctx.generate_u32(17 + t * 1000 + rnd, 1, n, stream=stream)
                    if (rnd + t) % 3 == 0:
                        keys &= 0x3FFFF  # two passes with a single digit
                    ref = keys.clone()
                    idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
                    stream.synchronize()
                    gate.wait()
                    in_b = False
                    for _ in range(args.sorts_per_round):
                        keys.copy_(ref)
                        in_b = plan.sort_pairs_indexed(keys, idx, kb, vb, stream=stream)
                    plan.check(stream=stream)
                    ok, ov = (kb, vb) if in_b else (keys, idx)
                    k64 = ok.to(torch.int64) & 0xFFFFFFFF
                    good = bool((k64[1:] >= k64[:-1]).all()) and bool((ref[ov.to(torch.int64)] == ok).all())
                    if not good:
                        bad[t] += 1
                        print(f"thread {t} round {rnd}: wrong output", flush=True)
                    if t == 0 and rnd % 10 == 9:
                        print(f"{rnd + 1} rounds", flush=True)
        except Exception as exc:  # noqa: BLE001
            errors.append((t, exc))
            gate.abort()

    threads = [threading.Thread(target=body, args=(t,)) for t in range(args.threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for t, exc in errors:
        if not isinstance(exc, threading.BrokenBarrierError):
            print(f"thread {t}: {exc}", flush=True)
    failed = bool(errors) or any(bad)
    total = args.threads * args.rounds * args.sorts_per_round
    print("CONCURRENT SOAK", "FAILED" if failed else "ok", f"({total} sorts of 2^{args.log2n} pairs from {args.threads} threads)")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
