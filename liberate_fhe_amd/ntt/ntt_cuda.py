"""`ntt_cuda`: the reference's 15-function extension module, served by libckks_hip.so.

Same names, argument order and mutation/allocation behaviour as the pybind11 module the reference
builds from src/liberate/ntt/ntt.cpp:421-437 — every tensor argument is a list with one entry per
participating GPU — so orchestration code written against the reference runs unchanged.  Each call
walks the lists and issues one C-ABI call per device on that device's current torch stream
(reference: ntt.cpp:130-141, K.cu:103-107).  Outputs of mont_mult / mont_add / mont_sub /
tile_unsigned are allocated here with torch (K.cu:138, 1209, 1217, 1225) so ownership stays with
torch's caching allocator.

Twiddles: the reference passes `even, odd, psi` = per-stage gather indices and a [rows, logN, N/2]
table.  The HIP kernels index a compact [rows, N] table.  `psi` may therefore be either
  * a 2-D compact table (what liberate_fhe_amd's own ntt_context passes; even/odd may be None), or
  * the reference's 3-D table, from which the compact table is extracted once and cached
    (entry m+i of stage s is column i*t of that stage, t = N/2m forward, t = 2^s inverse).
"""
from __future__ import annotations

import weakref

import torch

from .._native import lib, check
from . import twiddles

__all__ = [
    "mont_mult", "mont_enter", "ntt", "enter_ntt", "intt", "mont_redc", "intt_exit", "intt_exit_reduce",
    "intt_exit_reduce_signed", "reduce_2q", "make_signed", "make_unsigned", "mont_add", "mont_sub",
    "tile_unsigned",
]


def _dev_stream(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError(
            f"ntt_cuda: tensor on {t.device}; the HIP kernels need device memory (no CPU fallback)")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return idx, torch.cuda.current_stream(idx).cuda_stream


def _ptr(t: torch.Tensor):
    if t.dtype not in (torch.int64, torch.int32):
        raise TypeError(f"ntt_cuda: int64 tensors (62-bit word mode) or int32 tensors (30-bit word mode), got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("ntt_cuda: tensor must be contiguous")
    return t.data_ptr()


def _w30(t: torch.Tensor) -> bool:
    """The reference dispatches its kernel templates on the tensor's dtype (K.cu:141 AT_DISPATCH_INTEGRAL_TYPES): int32 =
    the 30-bit word mode (lf30_* entries), int64 = the 62-bit mode."""
    return t.dtype == torch.int32


def _same_words(what, a, *others):
    for o in others:
        if o is not None and o.dtype != a.dtype:
            raise TypeError(f"ntt_cuda.{what}: mixed word modes ({a.dtype} data with {o.dtype} constants)")


def _inplace(t: torch.Tensor):
    """Row-sliced views are contiguous and updated in place; anything else round-trips through a copy."""
    if t.is_contiguous():
        return t, None
    return t.contiguous(), t


_compact_cache = {}


def _compact(psi: torch.Tensor, inverse: bool) -> torch.Tensor:
    if psi.dim() == 2:
        return psi
    key = (id(psi), inverse)
    hit = _compact_cache.get(key)
    if hit is not None and hit[0]() is psi and hit[1] == psi._version:
        return hit[2]
    rows, logN, half = psi.shape
    N = 2 * half
    out = torch.zeros((rows, N), dtype=psi.dtype, device=psi.device)
    for s in range(logN):
        t = (1 << s) if inverse else (N >> (s + 1))
        m = (N >> (s + 1)) if inverse else (1 << s)
        out[:, m:2 * m] = psi[:, s, ::t]
    _compact_cache[key] = (weakref.ref(psi), psi._version, out)
    weakref.finalize(psi, _compact_cache.pop, key, None)
    return out


def mont_mult(a, b, ql, qh, kl, kh):
    out = []
    for ai, bi, l, h, kl_, kh_ in zip(a, b, ql, qh, kl, kh):
        dev, st = _dev_stream(ai)
        ai_c = ai.contiguous()
        bi_c = bi.contiguous()
        c = torch.empty_like(ai_c)
        _same_words("mont_mult", ai_c, bi_c, l, h, kl_, kh_)
        fn = lib.lf30_mont_mult if _w30(ai_c) else lib.lf_mont_mult
        check(fn(_ptr(ai_c), _ptr(bi_c), _ptr(c), ai_c.size(0), ai_c.size(-1),
                 _ptr(l), _ptr(h), _ptr(kl_), _ptr(kh_), dev, st), "mont_mult")
        out.append(c)
    return out


def mont_enter(a, Rs, ql, qh, kl, kh):
    for ai, r, l, h, kl_, kh_ in zip(a, Rs, ql, qh, kl, kh):
        dev, st = _dev_stream(ai)
        w, back = _inplace(ai)
        _same_words("mont_enter", w, r, l, h, kl_, kh_)
        fn = lib.lf30_mont_enter if _w30(w) else lib.lf_mont_enter
        check(fn(_ptr(w), _ptr(r.contiguous()), w.size(0), w.size(-1),
                 _ptr(l), _ptr(h), _ptr(kl_), _ptr(kh_), dev, st), "mont_enter")
        if back is not None:
            back.copy_(w)


def mont_redc(a, ql, qh, kl, kh):
    for ai, l, h, kl_, kh_ in zip(a, ql, qh, kl, kh):
        dev, st = _dev_stream(ai)
        w, back = _inplace(ai)
        _same_words("mont_redc", w, l, h, kl_, kh_)
        fn = lib.lf30_mont_redc if _w30(w) else lib.lf_mont_redc
        check(fn(_ptr(w), w.size(0), w.size(-1), _ptr(l), _ptr(h), _ptr(kl_), _ptr(kh_), dev, st),
              "mont_redc")
        if back is not None:
            back.copy_(w)


def _logN(ai):
    N = ai.size(-1)
    if N & (N - 1):
        raise ValueError("ntt_cuda: polynomial length must be a power of two")
    return N.bit_length() - 1


# Workspace of the two-launch transforms (lf_ntt_ws / lf_intt_ws, include/ckks_hip.h: 6-byte words between their passes), one per
# (device, stream) so that transforms on different streams never share it; grown on demand, as large as the largest stack
# transformed on that stream.  Set ntt_cuda.USE_WORKSPACE = False to transform strictly in place (lf_ntt / lf_intt), ~8 % slower on large stacks.
USE_WORKSPACE = True
_WS = {}


def _workspace(dev, st, rows, logN):
    if not USE_WORKSPACE or logN < 13 or logN > 17:
        return None
    words = int(lib.lf_ntt_ws_words(1, rows, logN))
    ws = _WS.get((dev, st))
    if ws is None or ws.numel() < words:
        ws = _WS[(dev, st)] = torch.empty((words,), dtype=torch.int64, device=f"cuda:{dev}")
    return ws


def _forward(a, Rs, psi, _2q, ql, qh, kl, kh, what):
    for i, ai in enumerate(a):
        dev, st = _dev_stream(ai)
        w, back = _inplace(ai)
        table = _compact(psi[i], inverse=False)
        rs = 0 if Rs is None else _ptr(Rs[i].contiguous())
        # one word mode per call, whichever it is: int64 data with int32 constants would be READ as int64 (wrong words, reads
        # past the end of the constants)
        _same_words(what, w, table, None if Rs is None else Rs[i], _2q[i], ql[i], qh[i], kl[i], kh[i])
        if _w30(w):   # 30-bit word mode: the plain per-stage transform (no fp64 class, no auxiliary table)
            check(lib.lf30_ntt(_ptr(w), 1, ql[i].size(0), _logN(w), _ptr(table), rs, _ptr(_2q[i]), _ptr(ql[i]), _ptr(qh[i]),
                               _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
            if back is not None:
                back.copy_(w)
            continue
        dp = twiddles.dp_pointer(table, ql[i], qh[i], kl[i], kh[i], dev, st)
        _, qhost = twiddles.host_primes(ql[i], qh[i])
        # extent = ql.size(0) rows (K.cu:298, 371)
        ws = _workspace(dev, st, ql[i].size(0), _logN(w))
        if ws is not None:
            check(lib.lf_ntt_ws(_ptr(w), _ptr(ws), 1, ql[i].size(0), _logN(w), _ptr(table), dp, qhost, rs, 0,
                                _ptr(ql[i]), _ptr(qh[i]), _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
        else:
            check(lib.lf_ntt(_ptr(w), 1, ql[i].size(0), _logN(w), _ptr(table), dp, qhost, rs, 0, _ptr(_2q[i]),
                             _ptr(ql[i]), _ptr(qh[i]), _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
        if back is not None:
            back.copy_(w)


def ntt(a, even, odd, psi, _2q, ql, qh, kl, kh):
    _forward(a, None, psi, _2q, ql, qh, kl, kh, "ntt")


def enter_ntt(a, Rs, even, odd, psi, _2q, ql, qh, kl, kh):
    _forward(a, Rs, psi, _2q, ql, qh, kl, kh, "enter_ntt")


def _inverse(a, psi, Ninv, _2q, ql, qh, kl, kh, tail, what):
    for i, ai in enumerate(a):
        dev, st = _dev_stream(ai)
        w, back = _inplace(ai)
        table = _compact(psi[i], inverse=True)
        _same_words(what, w, table, Ninv[i], _2q[i], ql[i], qh[i], kl[i], kh[i])
        if _w30(w):
            check(lib.lf30_intt(_ptr(w), 1, ql[i].size(0), _logN(w), _ptr(table), _ptr(Ninv[i].contiguous()), tail, _ptr(_2q[i]),
                                _ptr(ql[i]), _ptr(qh[i]), _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
            if back is not None:
                back.copy_(w)
            continue
        dp = twiddles.dp_pointer(table, ql[i], qh[i], kl[i], kh[i], dev, st)
        _, qhost = twiddles.host_primes(ql[i], qh[i])
        ws = _workspace(dev, st, ql[i].size(0), _logN(w))
        if ws is not None:
            check(lib.lf_intt_ws(_ptr(w), _ptr(ws), 1, ql[i].size(0), _logN(w), _ptr(table), dp, qhost, _ptr(Ninv[i].contiguous()), tail, 0,
                                 _ptr(ql[i]), _ptr(qh[i]), _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
        else:
            check(lib.lf_intt(_ptr(w), 1, ql[i].size(0), _logN(w), _ptr(table), dp, qhost, _ptr(Ninv[i].contiguous()), tail, 0,
                              _ptr(_2q[i]), _ptr(ql[i]), _ptr(qh[i]), _ptr(kl[i]), _ptr(kh[i]), dev, st), what)
        if back is not None:
            back.copy_(w)


def intt(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh):
    _inverse(a, psi, Ninv, _2q, ql, qh, kl, kh, 0, "intt")


def intt_exit(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh):
    _inverse(a, psi, Ninv, _2q, ql, qh, kl, kh, 1, "intt_exit")


def intt_exit_reduce(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh):
    _inverse(a, psi, Ninv, _2q, ql, qh, kl, kh, 2, "intt_exit_reduce")


def intt_exit_reduce_signed(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh):
    _inverse(a, psi, Ninv, _2q, ql, qh, kl, kh, 3, "intt_exit_reduce_signed")


def _fixup(fn, fn30, what):
    def op(a, _2q):
        for ai, q2 in zip(a, _2q):
            dev, st = _dev_stream(ai)
            w, back = _inplace(ai)
            _same_words(what, w, q2)
            check((fn30 if _w30(w) else fn)(_ptr(w), w.size(0), w.size(-1), _ptr(q2.contiguous()), dev, st), what)
            if back is not None:
                back.copy_(w)
    op.__name__ = what
    return op


reduce_2q = _fixup(lib.lf_reduce_2q, lib.lf30_reduce_2q, "reduce_2q")
make_signed = _fixup(lib.lf_make_signed, lib.lf30_make_signed, "make_signed")
make_unsigned = _fixup(lib.lf_make_unsigned, lib.lf30_make_unsigned, "make_unsigned")


def _binary(fn, fn30, what):
    def op(a, b, _2q):
        out = []
        for ai, bi, q2 in zip(a, b, _2q):
            dev, st = _dev_stream(ai)
            ai_c, bi_c = ai.contiguous(), bi.contiguous()
            c = torch.empty_like(ai_c)
            _same_words(what, ai_c, bi_c, q2)
            check((fn30 if _w30(ai_c) else fn)(_ptr(ai_c), _ptr(bi_c), _ptr(c), ai_c.size(0), ai_c.size(-1), _ptr(q2.contiguous()), dev, st), what)
            out.append(c)
        return out
    op.__name__ = what
    return op


mont_add = _binary(lib.lf_mont_add, lib.lf30_mont_add, "mont_add")
mont_sub = _binary(lib.lf_mont_sub, lib.lf30_mont_sub, "mont_sub")


def tile_unsigned(a, _2q):
    out = []
    for ai, q2 in zip(a, _2q):
        dev, st = _dev_stream(ai)
        ai.squeeze_()  # K.cu:1206
        src = ai.contiguous()
        c = src.new_empty((q2.size(0), src.size(0)))
        _same_words("tile_unsigned", src, q2)
        fn = lib.lf30_tile_unsigned if _w30(src) else lib.lf_tile_unsigned
        check(fn(_ptr(src), _ptr(c), q2.size(0), src.size(0), _ptr(q2.contiguous()), dev, st), "tile_unsigned")
        out.append(c)
    return out
