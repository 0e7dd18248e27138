"""Auxiliary twins of the compact twiddle tables: per limb 2N words — plain residues as doubles for the fp64
butterflies (primes below 2^41), (quotient, plain twiddle) Shoup pairs for the relaxed integer-class butterflies.

A table the kernels see may be a row-slice of a bigger per-device tensor (ntt_context hands out views per
level / per key-switch digit).  The fp64 copy therefore mirrors the whole underlying storage and rows are
filled on first use with `lf_twiddle_dp`, so every view of one table shares one copy.
"""
from __future__ import annotations

import weakref

import torch

from .._native import lib, check

_tables = {}   # id(base tensor) -> entry; entries die with their tensor (a freed tensor's address can be reused)
_views = {}    # id(view) -> (weakref, version, pointer, rows built, twin tensor): the per-launch fast path


def dp_pointer(table: torch.Tensor, ql, qh, kl, kh, dev: int, stream: int) -> int:
    """Device address of the fp64 twin of `table` ([rows, N] int64 Montgomery-form compact twiddles)."""
    hit = _views.get(id(table))
    if hit is not None and hit[0]() is table and hit[1] == table._version and hit[3] >= ql.size(0):
        return hit[2]                       # this view's rows are built: nothing to look at
    assert table.dim() == 2 and table.is_contiguous() and table.dtype == torch.int64
    N = table.size(1)
    base = table._base if table._base is not None else table
    key = id(base)
    entry = _tables.get(key)
    if entry is None or entry["ref"]() is not base or entry["version"] != base._version:
        # auxiliary rows are 2N words (fp64 class: N doubles; integer class: N (quotient, twiddle) Shoup pairs)
        entry = {"dp": torch.empty(2 * (base.untyped_storage().nbytes() // 8), dtype=torch.float64, device=table.device),
                 "built": set(), "ref": weakref.ref(base), "version": base._version}
        _tables[key] = entry
        weakref.finalize(base, _tables.pop, key, None)
    off = table.storage_offset()
    if off % N:
        raise ValueError("twiddle view must start on a row boundary of its storage")
    r0 = off // N
    rows = min(table.size(0), ql.size(0))
    missing = [r for r in range(rows) if (r0 + r) not in entry["built"]]
    if missing:
        lo, hi = missing[0], missing[-1] + 1
        check(lib.lf_twiddle_dp(table.data_ptr() + lo * N * 8, entry["dp"].data_ptr() + (r0 + lo) * N * 16, hi - lo, N,
                                ql.data_ptr() + lo * 8, qh.data_ptr() + lo * 8, kl.data_ptr() + lo * 8,
                                kh.data_ptr() + lo * 8, dev, stream), "lf_twiddle_dp")
        entry["built"].update(range(r0 + lo, r0 + hi))
        # The rows were filled by a launch on `stream`; the fast path above hands the address to launches on ANY
        # stream.  A build happens once per table and level, so it simply completes here (not per launch): afterwards
        # the rows are visible to every stream of the device.
        torch.cuda.ExternalStream(stream, device=dev).synchronize() if stream else torch.cuda.synchronize(dev)
    ptr = entry["dp"].data_ptr() + off * 16
    vkey = id(table)
    # the tuple holds the twin itself: the cached address cannot outlive its allocation
    _views[vkey] = (weakref.ref(table), table._version, ptr, rows, entry["dp"])
    weakref.finalize(table, _views.pop, vkey, None)
    return ptr


def twin_of(table: torch.Tensor) -> torch.Tensor:
    """The auxiliary tensor the last dp_pointer(table, ...) address points into (for callers that cache the raw address:
    holding the twin keeps that address valid)."""
    hit = _views.get(id(table))
    if hit is None or hit[0]() is not table:
        raise KeyError("dp_pointer() has not been called for this table")
    return hit[4]


_host_q = {}


def host_primes(ql: torch.Tensor, qh: torch.Tensor):
    """HOST int64 array of the primes behind device vectors ql/qh (one blocking copy per tensor pair, cached).
    Returns (numpy array kept alive by the cache, its address)."""
    key = (id(ql), id(qh))
    hit = _host_q.get(key)
    if hit is None or hit[0]() is not ql or hit[1]() is not qh or hit[2] != ql._version:
        q = ((qh.cpu() << 31) | ql.cpu()).numpy().copy()
        hit = (weakref.ref(ql), weakref.ref(qh), ql._version, q)
        _host_q[key] = hit
        weakref.finalize(ql, _host_q.pop, key, None)
    return hit[3], hit[3].ctypes.data
