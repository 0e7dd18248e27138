"""`randround_cuda`: the reference's extension module (src/liberate/csprng/randround.cpp) on libckks_hip.so."""
from __future__ import annotations

import torch

from .._native import lib, check
from ._dev import dev_stream, ptr

__all__ = ["randround"]


def randround(inputs, rand_bytes):
    """rand_bytes[i] (32-bit random words, int64) <- stochastic rounding of the fp64 inputs[i], in place."""
    for coef, rb in zip(inputs, rand_bytes):
        dev, st = dev_stream(rb, "randround")
        if coef.numel() != rb.numel():
            raise ValueError("randround: one random word per coefficient")
        check(lib.lf_randround(ptr(coef, "randround", torch.float64), ptr(rb, "randround"), rb.numel(), dev, st),
              "randround")
