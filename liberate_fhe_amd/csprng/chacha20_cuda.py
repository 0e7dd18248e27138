"""`chacha20_cuda`: the reference's extension module (src/liberate/csprng/chacha20.cpp:17-44) on libckks_hip.so."""
from __future__ import annotations

import torch

from .._native import lib, check
from ._dev import dev_stream, ptr

__all__ = ["chacha20"]


def chacha20(inputs, step):
    """inputs: one [n,16] int64 state table per GPU.  Returns the random blocks (new tensors); the
    inputs' block counters advance by `step`."""
    outputs = []
    for states in inputs:
        dev, st = dev_stream(states, "chacha20")
        dest = torch.empty_like(states)
        check(lib.lf_chacha20(ptr(states, "chacha20"), dest.data_ptr(), states.numel() // 16, int(step), dev, st),
              "chacha20")
        outputs.append(dest)
    return outputs
