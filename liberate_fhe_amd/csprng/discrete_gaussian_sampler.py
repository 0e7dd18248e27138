"""Cumulative-distribution table of the discrete Gaussian, laid out as an array binary tree.

Same table as the reference's `build_CDT_binary_search_tree`
(src/liberate/csprng/discrete_gaussian_sampler.py:12-118): half-plane probabilities of
exp(-x^2 / 2 sigma^2) / (sigma sqrt(2 pi)) at x = 0 .. 2^ceil(log2(6 sigma)) - 1 with the mass at 0
halved, accumulated at 2 x security_bits of working precision, scaled by 2^security_bits and
truncated; the cumulative values at the interior points 1 .. 2^depth - 1 are stored in level order
(root = midpoint) as (low 64 bits, high 64 bits).  tests/test_csprng_cpu.py compares the result
word for word with a golden copy of the reference's table.
"""
from __future__ import annotations

import math

import mpmath as mpm
import numpy as np

_keepalive = {}


def cumulative_table(security_bits=128, sigma=3.2):
    """The 2^depth + 1 cumulative integers (first entry 0, last ~ 2^(security_bits - 1))."""
    depth = math.ceil(math.log2(6 * sigma))
    with mpm.workprec(2 * security_bits):
        s = mpm.mpf(str(sigma))
        norm = s * mpm.sqrt(2 * mpm.pi)
        acc, table = mpm.mpf(0), [0]
        for x in range(2 ** depth):
            mass = mpm.exp(-mpm.mpf(x) ** 2 / (2 * s ** 2)) / norm
            acc = acc + (mass / 2 if x == 0 else mass)
            table.append(int(acc * mpm.mpf(2) ** security_bits))
    return table, depth


def level_order(depth):
    """Indices of the interior points, root first: depth d contributes the odd multiples of 2^(depth-1-d)."""
    order = []
    for d in range(depth):
        half = 2 ** (depth - 1 - d)
        order += list(range(half, 2 ** depth, 2 * half))
    return order


def build_CDT_binary_search_tree(security_bits=128, sigma=3.2):
    """Returns (btree uint64 [nodes, security_bits/64] low word first, host pointer to the flattened
    word-major copy the kernels take, nodes, depth)."""
    table, depth = cumulative_table(security_bits, sigma)
    words = security_bits // 64
    m64 = (1 << 64) - 1
    btree = np.array([[(table[i] >> (64 * w)) & m64 for w in range(words)] for i in level_order(depth)],
                     dtype=np.uint64)
    flat = np.ascontiguousarray(btree.T.ravel(), dtype=np.uint64)
    ptr = flat.__array_interface__["data"][0]
    _keepalive[ptr] = flat          # the pointer outlives this call
    return btree, ptr, btree.shape[0], depth
