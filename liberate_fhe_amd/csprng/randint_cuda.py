"""`randint_cuda`: the reference's extension module (src/liberate/csprng/randint.cpp:17-57) on libckks_hip.so.

`q_ptrs` are HOST addresses of contiguous uint64 arrays (one modulus per channel), as in the
reference (csprng.py:259-260); numpy arrays are accepted as well.
"""
from __future__ import annotations

import numpy as np
import torch

from .._native import lib, check
from ._dev import dev_stream, ptr

__all__ = ["randint", "randint_fast"]


def _host(q):
    if isinstance(q, np.ndarray):
        if q.dtype != np.uint64 or not q.flags.c_contiguous:
            raise TypeError("randint: moduli must be a contiguous uint64 array")
        return q.ctypes.data
    return int(q)


def randint(inputs, q_ptrs):
    """In place on random words [channels, n, 16]: word 4j of each row <- the sample of words 4j..4j+3."""
    for rb, q in zip(inputs, q_ptrs):
        dev, st = dev_stream(rb, "randint")
        check(lib.lf_randint(ptr(rb, "randint"), rb.size(0), rb.size(1), _host(q), dev, st), "randint")


def randint_fast(states, q_ptrs, shift, step):
    """states: one [channels, L, 16] table per GPU -> [channels, 4L] samples in [shift, q_c + shift)."""
    outputs = []
    for s, q in zip(states, q_ptrs):
        dev, st = dev_stream(s, "randint_fast")
        dst = s.new_empty((s.size(0), s.size(1) * 4))
        check(lib.lf_randint_fast(ptr(s, "randint_fast"), dst.data_ptr(), s.size(0), s.size(1), _host(q), int(shift),
                                  int(step), dev, st), "randint_fast")
        outputs.append(dst)
    return outputs
