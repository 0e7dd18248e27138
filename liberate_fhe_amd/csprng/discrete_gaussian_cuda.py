"""`discrete_gaussian_cuda`: the reference's extension module (src/liberate/csprng/discrete_gaussian.cpp)
on libckks_hip.so.  `btree_ptr` is the HOST address returned by build_CDT_binary_search_tree."""
from __future__ import annotations

from .._native import lib, check
from ._dev import dev_stream, ptr

__all__ = ["discrete_gaussian", "discrete_gaussian_fast"]


def discrete_gaussian(inputs, btree_ptr, btree_size, depth):
    """In place on random words [n,16]."""
    for rb in inputs:
        dev, st = dev_stream(rb, "discrete_gaussian")
        check(lib.lf_discrete_gaussian(ptr(rb, "discrete_gaussian"), rb.numel() // 16, int(btree_ptr), btree_size,
                                       depth, dev, st), "discrete_gaussian")


def discrete_gaussian_fast(states, btree_ptr, btree_size, depth, step):
    """states: one [n,16] table per GPU -> [4n] samples each."""
    outputs = []
    for s in states:
        dev, st = dev_stream(s, "discrete_gaussian_fast")
        n = s.numel() // 16
        dst = s.new_empty((n * 4,))
        check(lib.lf_discrete_gaussian_fast(ptr(s, "discrete_gaussian_fast"), dst.data_ptr(), n, int(btree_ptr),
                                            btree_size, depth, int(step), dev, st), "discrete_gaussian_fast")
        outputs.append(dst)
    return outputs
