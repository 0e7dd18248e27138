"""Test infrastructure: a host-staging transport underneath `torch.distributed.batch_isend_irecv`.

The product's exchanges (liberate_fhe_amd/fhe/comm.py) are batches of point-to-point messages on DEVICE buffers —
with RCCL they travel over xGMI.  A one-GPU box cannot run RCCL between two ranks (it refuses two ranks on one
device) and gloo moves host memory only, so the GPU rehearsals (tests/test_distributed_gpu.py, bench.py with
LF_BENCH_REHEARSE=1) install this transport: the product code builds and caches its P2POp list against the device
buffers exactly as it does for RCCL, calls dist.batch_isend_irecv, and the messages are carried by gloo through host
copies.  Nothing under liberate_fhe_amd/ imports this module.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class _StagedWork:
    def __init__(self, works, landings):
        self.works, self.landings = works, landings

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, host in self.landings:      # received rows: host -> their place in the device buffer, stream-ordered
            dst.copy_(host, non_blocking=False)
        self.landings = []


def install():
    """Idempotent.  Returns the list that records every batch: [(op name, peer, shape, is_cuda), ...] per call."""
    if getattr(dist.batch_isend_irecv, "_lf_staged", False):
        return dist.batch_isend_irecv._lf_log
    real = dist.batch_isend_irecv
    log = []

    def staged(ops):
        log.append([("send" if op.op is dist.isend else "recv", op.peer, tuple(op.tensor.shape), op.tensor.is_cuda) for op in ops])
        host_ops, landings = [], []
        for op in ops:
            t = op.tensor
            if not t.is_cuda:
                host_ops.append(op)
                continue
            if op.op is dist.isend:
                h = t.cpu()                  # blocks until the producing kernels on the current stream are done
            else:
                h = torch.empty(t.shape, dtype=t.dtype)
                landings.append((t, h))
            host_ops.append(dist.P2POp(op.op, h, op.peer, op.group))
        return [_StagedWork(real(host_ops), landings)]

    staged._lf_staged = True
    staged.moves_device_memory = True      # what DistComm looks for on a gloo group with device buffers
    staged._lf_log = log
    dist.batch_isend_irecv = staged
    return log
