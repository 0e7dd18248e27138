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


class _GatherWorks:
    """Handle of a padded all-gather: the wait orders the stream after the collective, then moves every foreign piece from
    the gathered slab to its rows of the storage-order buffer (device-to-device copies on the caller's stream)."""

    def __init__(self, work, moves):
        self.work, self.moves = work, moves

    def wait(self):
        if self.work is not None:
            self.work.wait()
        for dst, src in self.moves:
            dst.copy_(src)


class DistComm:
    """exchange = "p2p" (default): the key-switch digits travel as one batch of point-to-point messages between the ranks
    that hold rows; "allgather": as ONE padded all-gather (RCCL: ncclAllGather) over the whole group — the form SURVEY.md
    §8(e) and BASELINE's north star name.  Same words land in the same rows either way (tests/test_distributed_cpu.py);
    which one is faster on xGMI is a measurement (`bench.py --gpus N` times both)."""

    def __init__(self, group=None, local_device=None, exchange="p2p", solo_sharded=False):
        if not dist.is_initialized():
            raise RuntimeError("DistComm needs an initialised torch.distributed process group")
        if exchange not in ("p2p", "allgather"):
            raise ValueError("DistComm: exchange must be 'p2p' or 'allgather'")
        self.exchange = exchange
        # A group of ONE rank: the engine normally takes its single-device path and never calls the communicator.  solo_sharded
        # = True keeps the limb-sharded code path — the native halves around the two exchange steps, HIP-graph replay, and
        # every exchange issued to the backend (the all-gather form is a real collective of one rank; the point-to-point
        # batches are empty) — which is how a one-GPU lease exercises a real RCCL communicator underneath the engine.
        self.solo_sharded = bool(solo_sharded)
        self._slabs = {}
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        if self.solo_sharded and self.world_size != 1:
            raise ValueError("DistComm: solo_sharded is for a group of one rank")
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
    @property
    def whole_group_exchange(self):
        """True: exchange_rows is a collective of the WHOLE group — ranks without rows at the level call it too (buf None)."""
        return self.exchange == "allgather"

    def exchange_rows(self, buf, pieces, peers, width=None):
        """The digit exchange in this communicator's form (see the class docstring); `width` = words per row, needed
        by a rank that passes buf = None (whole-group form, no rows at the level)."""
        if self.exchange == "allgather":
            return self.exchange_rows_allgather(buf, pieces, peers, width)
        return self.exchange_rows_p2p(buf, pieces, peers)

    def exchange_rows_allgather(self, buf, pieces, peers, width=None):
        """The same exchange as ONE all-gather: every rank of the group contributes a slab of `most` rows — the pieces it
        owns packed one after the other, the rest padding (shards are unequal: gold over 8 GPUs owns 7 / 4 / .. / 4 rows) —
        and receives everybody's; the wait then copies every foreign piece to its rows of `buf`.  A collective of the whole
        group: ranks outside `peers` call it with buf = None, contribute padding and drop what they receive.  Costs one
        packing copy of the own rows, one scatter copy of the foreign ones, and `world x most` rows on the wire per rank
        against the exact rows of the point-to-point form."""
        world = self.world_size
        owned = {}
        for owner, row0, n in pieces:
            owned.setdefault(owner, []).append((row0, n))
        most = max(sum(n for _, n in v) for v in owned.values())
        if buf is not None:
            width, dtype, device = buf.shape[1], buf.dtype, buf.device
        else:
            if width is None:
                raise ValueError("exchange_rows_allgather: a rank without rows must pass the row width")
            dtype, device = torch.int64, torch.device(self.local_device)
        key = (most, width, dtype, str(device))
        slab = self._slabs.get(key)
        if slab is None:
            if len(self._slabs) > 64:
                self._slabs.clear()
            slab = self._slabs[key] = (torch.empty((most, width), dtype=dtype, device=device),
                                       torch.empty((world, most, width), dtype=dtype, device=device))
        send, recv = slab
        moves = []
        if buf is not None and self.rank in peers:
            at = 0
            for row0, n in owned.get(self.rank, []):
                send[at:at + n].copy_(buf[row0:row0 + n])
                at += n
            for owner, ps in owned.items():
                if owner == self.rank:
                    continue
                at = 0
                for row0, n in ps:
                    moves.append((buf[row0:row0 + n], recv[owner, at:at + n]))
                    at += n
        try:
            work = dist.all_gather_into_tensor(recv.view(world * most, width), send, group=self.group, async_op=True)
        except (RuntimeError, NotImplementedError):   # a backend without the flat form: the list form, same bytes
            work = dist.all_gather([recv[r] for r in range(world)], send, group=self.group, async_op=True)
        return _GatherWorks(work, moves)

    def exchange_rows_p2p(self, buf, pieces, peers):
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
