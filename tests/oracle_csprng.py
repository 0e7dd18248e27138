"""CPU stand-ins for the four sampler extension modules, built on oracle/csprng_oracle.py — the
checker's side of the sampler tests.

`make_modules()` returns modules with the reference extensions' function names and list-per-GPU calling
convention (chacha20.cpp:17, randint.cpp:21/38, discrete_gaussian.cpp, randround.cpp) operating on CPU
torch tensors.  They serve two purposes:
  * tests/golden/refdriver.py installs them as liberate.csprng.*_cuda, so the REFERENCE's own Csprng
    class (host logic, csprng.py) runs in the build container and defines the expected streams;
  * `OracleCsprng` = the product's Csprng host logic with these modules swapped in, for CPU tests of
    the host logic (stream layout, multi-rank invariants) and CPU engine round trips.
"""
from __future__ import annotations

import ctypes
import types

import numpy as np
import torch

from oracle import csprng_oracle as co


def _np(t):
    return t.numpy()


def _host_u64(ptr_or_array, count):
    if isinstance(ptr_or_array, np.ndarray):
        return np.asarray(ptr_or_array, dtype=np.uint64)
    buf = (ctypes.c_uint64 * count).from_address(int(ptr_or_array))
    return np.frombuffer(buf, dtype=np.uint64).copy()


def _stepped(states, fn):
    """Run fn on a contiguous copy of a (possibly sliced) state view and write the stepped counters back."""
    work = np.ascontiguousarray(_np(states)).copy()
    out = fn(work)
    _np(states)[...] = work
    return out


def make_modules():
    cha = types.ModuleType("chacha20_cuda")
    rint = types.ModuleType("randint_cuda")
    dg = types.ModuleType("discrete_gaussian_cuda")
    rr = types.ModuleType("randround_cuda")

    def chacha20(inputs, step):
        outs = []
        for s in inputs:
            def fn(w):
                blocks = co.chacha20_block(w.reshape(-1, 16))
                co.step_counter(w, step)
                return blocks
            outs.append(torch.from_numpy(_stepped(s, fn).reshape(tuple(s.shape))))
        return outs

    def randint_fast(states, q_ptrs, shift, step):
        return [torch.from_numpy(_stepped(s, lambda w, q=q, s=s: co.randint_fast(w, _host_u64(q, s.size(0)), shift, step)))
                for s, q in zip(states, q_ptrs)]

    def randint(inputs, q_ptrs):
        for rb, q in zip(inputs, q_ptrs):
            co.randint(_np(rb), _host_u64(q, rb.size(0)))

    def discrete_gaussian_fast(states, btree_ptr, btree_size, depth, step):
        tree = _host_u64(btree_ptr, 2 * btree_size)
        return [torch.from_numpy(_stepped(s, lambda w: co.discrete_gaussian_fast(w.reshape(-1, 16), tree, btree_size, depth, step)))
                for s in states]

    def discrete_gaussian(inputs, btree_ptr, btree_size, depth):
        tree = _host_u64(btree_ptr, 2 * btree_size)
        for rb in inputs:
            co.discrete_gaussian(_np(rb).reshape(-1, 16), tree, btree_size, depth)

    def randround(inputs, rand_bytes):
        for c, rb in zip(inputs, rand_bytes):
            co.randround(_np(c).ravel(), _np(rb).ravel())

    cha.chacha20 = chacha20
    rint.randint_fast, rint.randint = randint_fast, randint
    dg.discrete_gaussian_fast, dg.discrete_gaussian = discrete_gaussian_fast, discrete_gaussian
    rr.randround = randround
    return {"chacha20_cuda": cha, "randint_cuda": rint, "discrete_gaussian_cuda": dg, "randround_cuda": rr}


def oracle_csprng_class():
    from liberate_fhe_amd.csprng import Csprng
    mods = make_modules()
    return type("OracleCsprng", (Csprng,), dict(mods))
