"""Collectives used by the limb-sharded engine: one process per GPU, torch.distributed underneath.

On MI355X the backend is "nccl" (= RCCL) and the payloads travel over xGMI; the CPU test-suite runs the
same code over "gloo".  The path has exactly two exchange steps: a broadcast of the
rescaled-away limb row (2 x N words) and an all-gather of the key-switch digits (<= ceil(limbs/world)
rows per rank).  The reference stages both through pinned host memory
(src/liberate/fhe/ckks_engine.py:778-810, 999-1011).
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
