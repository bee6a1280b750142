"""Multi-GPU sort: shard by top-digit bucket, one process per GPU.

The reference is single-device; this is the scale-out of its sort.  Input is
block-distributed (rank r holds a contiguous slice, payload = global index).

  1. every rank histograms the top `split_bits` key bits of its slice;
  2. all-reduce(sum) of the 2^split_bits u64 counts -> global digit histogram (RCCL);
  3. every rank derives the same bucket -> rank map: contiguous bucket ranges whose
     cumulative counts are closest to multiples of N / world;
  4. one stable partition pass groups the local slice by destination rank;
  5. all-to-all-v of keys and payload (counts first), received in source-rank order;
  6. stable local sort of what arrived.  Rank r then holds the r-th slice of the
     global order, and because steps 4-6 are stable and slices arrive in rank order
     the global result equals a stable sort of the concatenated input.

The collectives are torch.distributed calls (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests); everything local goes through a
`LocalOps` object.  `HipLocalOps` is the product implementation (libmrs.so); the
CPU tests substitute a checker-backed double to exercise the orchestration.
"""
import numpy as np
import torch
import torch.distributed as dist


def bucket_to_rank_map(global_counts, world):
    """Deterministic bucket -> rank table from the global bucket histogram.

    Rank boundaries are placed after the bucket whose inclusive cumulative count is
    closest to r * N / world (ties to the lower bucket); the table is monotone.
    Returns a uint8 numpy array with one entry per bucket.
    """
    counts = np.asarray(global_counts, dtype=np.int64)
    nb = counts.size
    total = int(counts.sum())
    cum = np.cumsum(counts)
    cuts = []
    for r in range(1, world):
        target = (total * r) // world
        j = int(np.searchsorted(cum, target, side="left"))  # first bucket with cum >= target
        if j >= nb:
            cut = nb
        else:
            below = int(cum[j - 1]) if j > 0 else 0
            # buckets [0, j) hold `below`, buckets [0, j] hold cum[j]
            cut = j if (target - below) <= (int(cum[j]) - target) else j + 1
        cuts.append(cut)
    cuts = np.maximum.accumulate(np.asarray(cuts, dtype=np.int64)) if cuts else np.zeros(0, dtype=np.int64)
    table = np.searchsorted(cuts, np.arange(nb), side="right").astype(np.uint8)
    return table


class HipLocalOps:
    """Local stages on one MI355X through the C ABI."""

    def __init__(self, ctx, capacity, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=0, end_bit=None):
        from .api import ALGO_ONESWEEP, SortPlan

        self.ctx = ctx
        self.device = ctx.device
        self.key_words = key_words
        self.plan = SortPlan(ctx, capacity, key_words=key_words, payload_bytes=payload_bytes, digit_bits=digit_bits,
                             begin_bit=begin_bit, end_bit=end_bit, algo=ALGO_ONESWEEP)

    def empty(self, n, words):
        return torch.empty(max(1, n * words), dtype=torch.int32, device=self.device)

    def bucket_histogram(self, keys, n, shift, bits):
        return self.ctx.bucket_histogram(keys, n, self.key_words, shift, bits)

    def partition(self, keys, vals, n, keys_out, vals_out, shift, bits, table, parts):
        table_dev = torch.from_numpy(table).to(self.device)
        self.plan.partition(keys, vals, keys_out, vals_out, shift, bits, table_dev, parts, n=n)

    def sort(self, keys_a, vals_a, keys_b, vals_b, n):
        in_b = self.plan.sort_pairs(keys_a, vals_a, keys_b, vals_b, n=n)
        return (keys_b, vals_b) if in_b else (keys_a, vals_a)

    def mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @staticmethod
    def elapsed_ms(a, b):
        torch.cuda.synchronize()
        return a.elapsed_time(b)


class DistributedSorter:
    def __init__(self, ctx, n_local, world, rank, digit_bits=8, split_bits=12, key_words=1, payload_words=1,
                 end_bit=None, slack=1.3, ops=None, group=None):
        if world > 16:
            raise ValueError("at most 16 ranks (one partition pass with 16 destinations)")
        self.world, self.rank, self.group = world, rank, group
        self.n_local = n_local
        self.key_words, self.payload_words = key_words, payload_words
        self.end_bit = 32 * key_words if end_bit is None else end_bit
        self.split_bits = min(split_bits, self.end_bit)
        self.shift = self.end_bit - self.split_bits
        self.capacity = int(n_local * slack) + 1024
        self.ops = ops or HipLocalOps(ctx, self.capacity, key_words=key_words, payload_bytes=4 * payload_words,
                                      digit_bits=digit_bits, end_bit=self.end_bit)
        o = self.ops
        self.part_k, self.part_v = o.empty(n_local, key_words), o.empty(n_local, payload_words)
        self.recv_k, self.recv_v = o.empty(self.capacity, key_words), o.empty(self.capacity, payload_words)
        self.tmp_k, self.tmp_v = o.empty(self.capacity, key_words), o.empty(self.capacity, payload_words)
        self._marks = None
        self.n_recv = 0
        self.table = None

    def sort(self, keys, vals, n=None):
        """Sorts the distributed array in place of its slices; returns (keys, vals) views of
        length self.n_recv holding this rank's slice of the global order."""
        o, P = self.ops, self.world
        n = self.n_local if n is None else n
        kw, pw = self.key_words, self.payload_words
        m0 = o.mark()
        # 1-2. global digit histogram
        local_counts = o.bucket_histogram(keys, n, self.shift, self.split_bits)
        global_counts = local_counts.clone()
        dist.all_reduce(global_counts, op=dist.ReduceOp.SUM, group=self.group)
        # 3. bucket -> rank table (same on every rank)
        table = bucket_to_rank_map(global_counts.cpu().numpy(), P)
        self.table = table
        send_counts = np.bincount(table, weights=local_counts.cpu().numpy().astype(np.float64), minlength=P)
        send_counts = send_counts.astype(np.int64)
        m1 = o.mark()
        # 4. stable partition by destination
        o.partition(keys, vals, n, self.part_k, self.part_v, self.shift, self.split_bits, table, P)
        m2 = o.mark()
        # 5. exchange: counts, then keys and payload, received in source-rank order
        send_t = torch.from_numpy(send_counts).to(local_counts.device)
        recv_t = torch.empty_like(send_t)
        dist.all_to_all_single(recv_t, send_t, group=self.group)
        recv_counts = recv_t.cpu().numpy()
        n_recv = int(recv_counts.sum())
        if n_recv > self.capacity:
            raise RuntimeError(f"rank {self.rank}: {n_recv} elements arrive but the receive buffers hold {self.capacity}; "
                               "the top-bit buckets are too unbalanced for this slack")
        self.n_recv = n_recv
        for src, dst, words in ((self.part_k, self.recv_k, kw), (self.part_v, self.recv_v, pw)):
            dist.all_to_all_single(dst[: n_recv * words], src[: n * words],
                                   output_split_sizes=[int(c) * words for c in recv_counts],
                                   input_split_sizes=[int(c) * words for c in send_counts], group=self.group)
        m3 = o.mark()
        # 6. local sort
        out_k, out_v = o.sort(self.recv_k, self.recv_v, self.tmp_k, self.tmp_v, n_recv)
        m4 = o.mark()
        self._marks = (m0, m1, m2, m3, m4)
        return out_k[: n_recv * kw], out_v[: n_recv * pw]

    def last_breakdown(self):
        """Phase times of the last sort() on this rank, in milliseconds."""
        if not self._marks:
            return {}
        m, e = self._marks, self.ops.elapsed_ms
        return {
            "dist_histogram_allreduce_ms": e(m[0], m[1]),
            "dist_partition_ms": e(m[1], m[2]),
            "dist_exchange_ms": e(m[2], m[3]),
            "dist_local_sort_ms": e(m[3], m[4]),
            "dist_recv_pairs_rank0": self.n_recv,
        }

    def verify(self, out_k, out_v):
        """Global checks without gathering the data: slices sorted, slice boundaries ordered,
        element count and key checksum conserved.  32-bit keys only."""
        k = out_k.to(torch.int64) & 0xFFFFFFFF
        ok = bool((k[1:] >= k[:-1]).all()) if k.numel() > 1 else True
        first = int(k[0]) if k.numel() else -1
        last = int(k[-1]) if k.numel() else -1
        info = torch.tensor([first, last, k.numel(), int(k.sum()), int(ok)], dtype=torch.int64, device=out_k.device)
        gathered = [torch.empty_like(info) for _ in range(self.world)]
        dist.all_gather(gathered, info, group=self.group)
        rows = [g.cpu().numpy() for g in gathered]
        good = all(int(r[4]) == 1 for r in rows)
        prev_last = -1
        for r in rows:
            if int(r[2]) == 0:
                continue
            good = good and prev_last <= int(r[0])
            prev_last = int(r[1])
        good = good and sum(int(r[2]) for r in rows) == self.n_local * self.world
        return good
