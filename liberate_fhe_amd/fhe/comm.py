"""Collectives used by the limb-sharded engine: one process per GPU, torch.distributed underneath.

On MI355X the backend is "nccl" (= RCCL) and the payloads travel over xGMI; the CPU test-suite runs the
same code over "gloo".  The path has exactly two exchange steps, both IN PLACE on buffers the
engine allocates once per level (`broadcast_into`):

  * rescale: the dropped limb's two rows (2 x N words) from their owner to every rank;
  * key switch: every Garner digit from its owner to every rank, ONE collective per run of digits with the same
    owner, each straight into its rows of the storage-order digit buffer — shards of unequal height travel
    unpadded, nothing is concatenated or re-indexed afterwards.  The collectives are issued asynchronously, in
    consumption order, before any extension starts; the engine waits on a group's handle only when it launches
    that group's extension + NTT (lf_ks_fwd), so group g + 1 is on the wire while the kernels of group g run.
    With RCCL the wait is a stream dependency (the communicator's stream -> the compute stream), not a host
    block.  xGMI is point to point: a digit of 4 limbs at gold is 2 MiB = ~15 us on one 153 GB/s link, and the
    owners of consecutive groups are different GPUs, so consecutive broadcasts leave over different links.

The reference stages both exchanges through pinned host memory (src/liberate/fhe/ckks_engine.py:778-810,
999-1011) and starts extending only when every digit has landed on every GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DistComm:
    def __init__(self, group=None, local_device=None):
        if not dist.is_initialized():
            raise RuntimeError("DistComm needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.local_device = local_device if local_device is not None else (
            f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu")

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

    def broadcast_into(self, buf, src, async_op=False):
        """In-place broadcast of the preallocated, contiguous `buf` (payload on group rank `src`, destination
        elsewhere).  async_op: returns the work handle; `.wait()` orders the caller's current stream after it."""
        if not buf.is_contiguous():
            raise ValueError("broadcast_into needs a contiguous buffer (a row range of a [rows, N] tensor is)")
        return dist.broadcast(buf, src=self._global(src), group=self.group, async_op=async_op)

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
