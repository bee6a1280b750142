"""The sort sharded over the GPUs of a node: host mirror of `mrs_comm_*` / `mrs_dist_*`.

The reference is single-device; this is the scale-out of its sort (shard by top-digit bucket).
Everything — histograms, RCCL collectives, the splitter, partition, exchange, local sort — runs
inside libmrs.so (`csrc/mrs_dist.hip`); this module only creates the handles and wraps the
library-owned result memory as torch tensors.  torch.distributed is used for one thing: handing
rank 0's RCCL unique id to the other ranks.
"""
import ctypes

import numpy as np
import torch

from . import _lib

UNIQUE_ID_BYTES = 128


class Communicator:
    """`mrs_comm`: the transport of the sharded sort."""

    def __init__(self, handle, rank, world, lib):
        self._h, self.rank, self.world, self.lib = handle, rank, world, lib

    @classmethod
    def rccl(cls, rank, world, group=None):
        """One rank per process, RCCL underneath.  Collective over `group` (torch.distributed),
        which only carries the 128-byte unique id from rank 0 to the others."""
        import torch.distributed as dist

        lib = _lib.load_library()
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        if rank == 0:
            _lib.check(lib.mrs_comm_get_unique_id(buf))
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = ctypes.create_string_buffer(box[0], UNIQUE_ID_BYTES)
        h = ctypes.c_void_p()
        _lib.check(lib.mrs_comm_create_rccl(buf, rank, world, ctypes.byref(h)))
        return cls(h, rank, world, lib)

    @classmethod
    def local(cls, world):
        """`world` ranks inside this process (one host thread per rank must drive them)."""
        lib = _lib.load_library()
        arr = (ctypes.c_void_p * world)()
        _lib.check(lib.mrs_comm_create_local(world, arr))
        return [cls(ctypes.c_void_p(arr[r]), r, world, lib) for r in range(world)]

    def close(self):
        if self._h:
            self.lib.mrs_comm_destroy(self._h)
            self._h = None


class _DeviceView:
    """A library-owned device buffer presented through the CUDA array interface."""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def _as_tensor(ptr, words, device):
    if words == 0 or not ptr:
        return torch.empty(0, dtype=torch.int32, device=device)
    return torch.as_tensor(_DeviceView(ptr, words), device=device)


_SCHEDULE_NAMES = {0: "passes", 1: "msd", 2: "msd refused by the device", 3: "msd adaptive"}


class DistributedSorter:
    """`mrs_dist`: rank `comm.rank`'s end of the sharded sort of (key, payload) pairs."""

    def __init__(self, ctx, comm, n_local_max, key_words=1, payload_words=1, end_bit=None, tolerance=0, exchange_chunks=0):
        self.ctx, self.comm, self.lib = ctx, comm, ctx.lib
        self.key_words, self.payload_words = key_words, payload_words
        self.end_bit = 32 * key_words if end_bit is None else end_bit
        desc = _lib.DistDesc(n_local_max, key_words, 4 * payload_words, self.end_bit, exchange_chunks, tolerance)
        h = ctypes.c_void_p()
        _lib.check(self.lib.mrs_dist_create(ctx._h, comm._h, ctypes.byref(desc), ctypes.byref(h)))
        self._h = h
        self.n_local_max = n_local_max
        self.n_out = 0

    def close(self):
        if self._h:
            self.lib.mrs_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        raw = _lib.DistInfoStruct()
        _lib.check(self.lib.mrs_dist_get_info(self._h, ctypes.byref(raw)))
        return raw

    def set_timing(self, enable=True):
        _lib.check(self.lib.mrs_dist_set_timing(self._h, 1 if enable else 0))

    def sort(self, keys, vals, n=None, stream=None):
        """Collective.  Returns (keys, vals) tensors (views of memory the library owns, valid until
        the next call) holding this rank's slice of the global order; `self.n_out` is its length."""
        n = (keys.numel() // self.key_words) if n is None else n
        if stream is None:
            stream = torch.cuda.current_stream()
        ko, vo, n_out = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        kp = ctypes.c_void_p(keys.data_ptr()) if n else None
        vp = ctypes.c_void_p(vals.data_ptr()) if (n and self.payload_words) else None
        _lib.check(self.lib.mrs_dist_sort_pairs(self._h, n, kp, vp, ctypes.byref(ko), ctypes.byref(vo),
                                                ctypes.byref(n_out), ctypes.c_void_p(stream.cuda_stream)))
        self.n_out = int(n_out.value)
        dev = self.ctx.device
        out_k = _as_tensor(ko.value, self.n_out * self.key_words, dev)
        out_v = _as_tensor(vo.value, self.n_out * self.payload_words, dev) if self.payload_words else None
        return out_k, out_v

    def check(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream()
        _lib.check(self.lib.mrs_dist_check(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def last_breakdown(self):
        """Phase times of the last sort() on this rank in milliseconds (timing must be on)."""
        i = self.info()
        return {
            "dist_histogram_collectives_split_ms": float(i.last_ms[0]),
            "dist_partition_ms": float(i.last_ms[1]),
            "dist_exchange_ms": float(i.last_ms[2]),
            "dist_local_sort_ms": float(i.last_ms[3]),
            "dist_recv_pairs_rank0": int(i.last_n_out),
            "dist_splitter_levels": int(i.last_levels),
            "dist_exact_cuts": int(i.last_exact_cuts),
            "dist_local_sort_pass_ms": float(i.last_sort_pass_ms),
            "dist_local_sort_passes": int(i.last_sort_passes),
            "dist_local_sort_piece_pairs": int(i.last_sort_piece_pairs),
            "dist_local_sort_schedule": [_SCHEDULE_NAMES.get(int(x), str(int(x))) for x in i.last_sort_schedule],
        }


def pair_checksum(keys, vals):
    """Order-independent 64-bit checksum of (key, payload) pairs (32-bit keys), on the device."""
    k = keys.to(torch.int64) & 0xFFFFFFFF
    v = vals.to(torch.int64) & 0xFFFFFFFF
    h = (k * 0x9E3779B1 + v * 0x85EBCA77 + ((k << 32) ^ v) * 0x27D4EB2F) ^ (k * v)
    return int(h.sum())  # int64 arithmetic wraps: a sum modulo 2^64


def verify_distributed(out_k, out_v, checksum_in, n_in, group=None):
    """Global checks without gathering the data (32-bit keys, payload = global element index,
    collective): every slice sorted with ties in payload order, slice boundaries ordered the same
    way, element count conserved and the multiset of (key, payload) pairs conserved (checksum)."""
    import torch.distributed as dist

    k = out_k.to(torch.int64) & 0xFFFFFFFF
    v = out_v.to(torch.int64) & 0xFFFFFFFF
    ok = True
    if k.numel() > 1:
        dk = k[1:] - k[:-1]
        ok = bool(((dk > 0) | ((dk == 0) & (v[1:] > v[:-1]))).all())
    first = (int(k[0]), int(v[0])) if k.numel() else (-1, -1)
    last = (int(k[-1]), int(v[-1])) if k.numel() else (-1, -1)
    wrap = lambda x: x - (1 << 64) if x >= (1 << 63) else x
    info = torch.tensor([first[0], first[1], last[0], last[1], k.numel(), wrap(pair_checksum(out_k, out_v) % (1 << 64)),
                         wrap(checksum_in % (1 << 64)), n_in, int(ok)], dtype=torch.int64, device=out_k.device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        gathered = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(gathered, info, group=group)
        rows = [g.cpu().numpy() for g in gathered]
    else:
        rows = [info.cpu().numpy()]
    good = all(int(r[8]) == 1 for r in rows)
    prev = None
    for r in rows:
        if int(r[4]) == 0:
            continue
        if prev is not None:
            good = good and (prev < (int(r[0]), int(r[1])))
        prev = (int(r[2]), int(r[3]))
    good = good and sum(int(r[4]) for r in rows) == sum(int(r[7]) for r in rows)
    mask = (1 << 64) - 1
    good = good and (sum(int(r[5]) for r in rows) & mask) == (sum(int(r[6]) for r in rows) & mask)
    return good
