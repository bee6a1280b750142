#!/usr/bin/env python3
"""One sharded sort with a single in-process rank (or one RCCL rank with --rccl) at a chosen size,
checked element by element against torch's stable sort: says where the output first differs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_morton_code_radix_sort_test_amd as mrs  # noqa: E402
from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--rccl", action="store_true")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--watchdog", type=int, default=100, help="seconds a multi-rank iteration may take")
    ap.add_argument("--world", type=int, default=1,
                    help="in-process ranks on GPU 0 (checked through the properties of one global stable sort)")
    args = ap.parse_args()
    if args.world > 1:
        sys.exit(many_ranks(args))
    n = args.n or (1 << args.log2n)
    torch.cuda.set_device(0)
    ctx = mrs.Context(0)
    if args.rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        comm = Communicator.rccl(0, 1)
    else:
        comm = Communicator.local(1)[0]
    sorter = DistributedSorter(ctx, comm, n, key_words=1, payload_words=1, end_bit=30, exchange_chunks=args.chunks)
    bad = False
    for it in range(args.iters):
        xyz = ctx.generate_points(77 + it, n)
        k, i = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
        del xyz
        ok, ov = sorter.sort(k, i)
        sorter.check()
        torch.cuda.synchronize()
        print(f"iteration {it}: {ok.numel()} of {n} pairs came back", flush=True)
        ref_k, perm = torch.sort(k.to(torch.int64), stable=True)
        ref_v = i[perm]
        del perm
        same_k = ok.to(torch.int64) == ref_k
        same_v = ov == ref_v
        nk, nv = int((~same_k).sum()), int((~same_v).sum())
        print(f"  keys differing: {nk}, payloads differing: {nv}", flush=True)
        if nk or nv:
            bad = True
            w = torch.nonzero(~(same_k & same_v)).flatten()
            print(f"  first at {int(w[0])}, last at {int(w[-1])}", flush=True)
            for j in w[:4].tolist() + w[-2:].tolist():
                print(f"    [{j}] got ({int(ok[j]) & 0xFFFFFFFF:#x}, {int(ov[j])}) want ({int(ref_k[j]):#x}, {int(ref_v[j])})")
            # how the differing positions spread
            h = torch.histc(w.to(torch.float64), bins=16, min=0, max=float(n))
            print("  per sixteenth of the output:", [int(x) for x in h.tolist()], flush=True)
        del ref_k, ref_v, same_k, same_v
    print("DIST DIAG", "FAILED" if bad else "ok")
    sys.exit(1 if bad else 0)


def many_ranks(args):
    import threading

    from wgpu_morton_code_radix_sort_test_amd.dist import pair_checksum

    n, world = args.n or (1 << args.log2n), args.world
    comms = Communicator.local(world)
    bad = False
    import faulthandler

    for it in range(args.iters):
        outs, sums, errors = [None] * world, [0] * world, []
        torch.cuda.empty_cache()  # the library allocates beside torch's cache
        faulthandler.dump_traceback_later(args.watchdog, exit=True)  # a stuck rank: say where every thread is

        def body(r):
            try:
                torch.cuda.set_device(0)
                c = mrs.Context(0)
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    xyz = c.generate_points(77 + 10 * it + r, n, stream=stream)
                    k, i = c.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], stream=stream)
                    del xyz
                    i += r * n
                    sums[r] = pair_checksum(k, i)
                    print(f"  rank {r}: input made", flush=True)
                    sorter = DistributedSorter(c, comms[r], n, key_words=1, payload_words=1, end_bit=30,
                                               exchange_chunks=args.chunks)
                    ok, ov = sorter.sort(k, i, stream=stream)
                    sorter.check(stream=stream)
                    stream.synchronize()
                    print(f"  rank {r}: sorted", flush=True)
                    outs[r] = (ok.clone(), ov.clone(), [int(x) for x in sorter.info().last_sort_schedule])
                    sorter.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((r, exc))
                print(f"  rank {r} gave up: {exc}", flush=True)
                comms[r].close()  # the other ranks must not wait for this one

        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        faulthandler.cancel_dump_traceback_later()
        for r, exc in errors:
            print(f"rank {r}: {exc}", flush=True)
        if errors:
            return 1
        total, check, prev = 0, 0, None
        for r in range(world):
            ok, ov, sched = outs[r]
            k = ok.to(torch.int64) & 0xFFFFFFFF
            v = ov.to(torch.int64) & 0xFFFFFFFF
            dk = k[1:] - k[:-1]
            good = bool(((dk > 0) | ((dk == 0) & (v[1:] > v[:-1]))).all())
            if k.numel():
                first, last = (int(k[0]), int(v[0])), (int(k[-1]), int(v[-1]))
                if prev is not None and not prev < first:
                    good = False
                prev = last
            total += k.numel()
            check = (check + pair_checksum(ok, ov)) & ((1 << 64) - 1)
            print(f"iteration {it} rank {r}: {k.numel()} pairs, in order: {good}, schedules {sched}", flush=True)
            bad = bad or not good
            del k, v, dk, ok, ov
        del outs
        want = sum(sums) & ((1 << 64) - 1)
        print(f"iteration {it}: {total} of {world * n} pairs, checksum {'kept' if check == want else 'CHANGED'}", flush=True)
        bad = bad or total != world * n or check != want
    [c.close() for c in comms]
    print("DIST DIAG", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    main()
