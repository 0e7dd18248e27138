"""Exchanges of the limb-sharded engine: one process per GPU, torch.distributed underneath.

On MI355X the backend is "nccl" (= RCCL) and the payloads travel over xGMI; the CPU test-suite runs the
same code over "gloo".  The path has exactly two exchange steps, both IN PLACE on buffers the
engine allocates once per level:

  * key switch (`exchange_rows`): every rank needs every Garner digit.  xGMI is point to point — each GPU has its own
    link to each of the other seven — so the digits travel as ONE batch of point-to-point messages (RCCL: one
    ncclGroup of send / recv pairs): a rank sends the runs of digits it owns straight to every other alive rank and
    receives every other run from its owner, each into its rows of the storage-order digit buffer.  Every link carries
    only its own pair's traffic (gold, 8 GPUs: ~2.5 MB per link, all links at once) instead of ten ring collectives
    of 2 MiB queued one behind the other on the communicator's stream.  Shards of unequal height travel unpadded,
    nothing is concatenated or re-indexed afterwards, ranks without rows at the level take no part.  The batch is
    asynchronous: the engine extends + transforms the digits it OWNS (lf_ks_fwd) while the others are on the wire,
    then orders its stream after the batch (with RCCL a stream dependency, not a host block) and does the rest.
  * rescale (`fanout_into`): the dropped limb's rows of all operands, one message from their owner to each rank that
    still holds rows at the next level.

The reference stages both exchanges through pinned host memory (src/liberate/fhe/ckks_engine.py:778-810,
999-1011) and starts extending only when every digit has landed on every GPU.
"""
from __future__ import annotations

import weakref

import torch
import torch.distributed as dist


class _Works:
    """The handles of one batch of messages."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class DistComm:
    def __init__(self, group=None, local_device=None):
        if not dist.is_initialized():
            raise RuntimeError("DistComm needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self._ops = {}
        self.local_device = local_device if local_device is not None else (
            f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu")
        self.backend_name = str(dist.get_backend(group))
        # gloo moves host memory only: device buffers need RCCL ("nccl") — or, on a one-GPU rehearsal box, a transport
        # underneath dist.batch_isend_irecv that declares it stages device memory (tests/gloo_device_p2p.py).  Fail HERE
        # with the reason, not deep inside the first key switch with the backend's own message.
        if str(self.local_device).startswith("cuda") and "gloo" in self.backend_name and "nccl" not in self.backend_name \
                and not getattr(dist.batch_isend_irecv, "moves_device_memory", False):
            raise RuntimeError("DistComm: the process group's backend is gloo, which cannot move the engine's DEVICE buffers "
                               "point to point; initialise torch.distributed with backend 'nccl' (RCCL) for GPU ranks")

    def _global(self, group_rank):
        """torch.distributed addresses peers by GLOBAL rank; the engine speaks in ranks of its group."""
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def broadcast(self, tensor, src, shape, device):
        """`tensor` is the payload on (group) rank `src` and ignored (may be None) elsewhere."""
        if self.rank != src:
            tensor = torch.empty(shape, dtype=torch.int64, device=device)
        else:
            tensor = tensor.contiguous()
        dist.broadcast(tensor, src=self._global(src), group=self.group)
        return tensor

    # ---- point-to-point exchanges ---------------------------------------------------------------------------------
    # Device buffers need a backend that moves device memory point to point: RCCL ("nccl").  gloo carries host tensors
    # only — the CPU test-suite runs this very code over it; a GPU rehearsal over gloo installs the host-staging
    # transport of tests/gloo_device_p2p.py underneath dist.batch_isend_irecv (test infrastructure, not a fallback).
    def exchange_rows(self, buf, pieces, peers):
        """All-pairs exchange, in place on the contiguous [rows, N] `buf`.  pieces = [(owner, first row, rows), ..]
        (group ranks; every rank passes the same list), peers = the group ranks taking part.  This rank sends the
        pieces it owns to every other peer and receives every other piece from its owner — one batch of asynchronous
        point-to-point messages.  Returns a handle; `.wait()` orders the caller's current stream after the batch.
        Ranks outside `peers` (no rows at the level) neither send nor receive: nothing here is a whole-group collective."""
        if self.rank not in peers:
            return _Works([])
        # the message list of a (buffer, schedule) pair is built once: the engine's buffers and schedules live as long
        # as the engine does, and the key switch is called thousands of times per second
        key = (buf.data_ptr(), tuple(buf.shape), tuple(pieces), tuple(peers))
        hit = self._ops.get(key)
        if hit is not None and hit[0]() is not buf:
            hit = None      # the address was recycled by another tensor: the cached messages point at freed storage
        if hit is None:
            ops = []
            for owner, row0, n in pieces:
                part = buf[row0:row0 + n]
                if owner == self.rank:
                    ops += [dist.P2POp(dist.isend, part, self._global(p), self.group) for p in peers if p != self.rank]
                else:
                    ops.append(dist.P2POp(dist.irecv, part, self._global(owner), self.group))
            if len(self._ops) > 256:
                self._ops.clear()
            hit = self._ops[key] = (weakref.ref(buf), ops)
        ops = hit[1]
        return _Works(dist.batch_isend_irecv(ops) if ops else [])

    def fanout_into(self, buf, src, peers):
        """The contiguous `buf` from group rank `src` to every rank of `peers` (in place, blocking the stream only)."""
        if self.rank == src:
            ops = [dist.P2POp(dist.isend, buf, self._global(p), self.group) for p in peers if p != src]
        elif self.rank in peers:
            ops = [dist.P2POp(dist.irecv, buf, self._global(src), self.group)]
        else:
            ops = []
        _Works(dist.batch_isend_irecv(ops) if ops else []).wait()

    def all_gather(self, tensor):
        out = [torch.empty_like(tensor) for _ in range(self.world_size)]
        dist.all_gather(out, tensor.contiguous(), group=self.group)
        return out

    def broadcast_int(self, value: int, src: int = 0) -> int:
        t = torch.tensor([value], dtype=torch.int64, device=self.local_device)
        dist.broadcast(t, src=self._global(src), group=self.group)
        return int(t.item())

    def barrier(self):
        dist.barrier(group=self.group)
