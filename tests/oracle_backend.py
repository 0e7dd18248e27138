"""CPU checker backend: the engine's backend interface implemented with the C oracle.

TEST INFRASTRUCTURE.  Gives the test-suite two things:
  * `make_ops()`   — the 15 `ntt_cuda` functions (reference signatures, list-of-tensors) on CPU tensors,
                     used (a) as the stand-in under the REAL reference engine when golden vectors are
                     generated and (b) under this repo's ntt_context in CPU tests;
  * `OracleBackend`— every fused engine op restated as the SEQUENCE of reference primitives it replaces
                     (citing src/liberate/fhe/ckks_engine.py), so the HIP engine can be compared with an
                     independent composition on any seeded input, and the multi-rank orchestration can
                     run on CPU over gloo.
Never imported by the product package.
"""
from __future__ import annotations

import types

import numpy as np
import torch

from oracle import oracle as orc


def _np(t):
    assert t.dtype in (torch.int64, torch.int32) and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return t.numpy()


def _inplace(t):
    if t.is_contiguous():
        return t, None
    return t.contiguous(), t


def make_ops(name="oracle_ntt_cuda"):
    m = types.ModuleType(name)

    def mont_mult(a, b, ql, qh, kl, kh):
        out = []
        for ai, bi, l, h, kl_, kh_ in zip(a, b, ql, qh, kl, kh):
            ai_c = ai.contiguous()
            c = torch.empty_like(ai_c)
            orc.mont_mult(_np(ai_c), _np(bi.contiguous()), _np(c), ai_c.size(0), _np(l), _np(h), _np(kl_), _np(kh_))
            out.append(c)
        return out

    def mont_enter(a, Rs, ql, qh, kl, kh):
        for ai, r, l, h, kl_, kh_ in zip(a, Rs, ql, qh, kl, kh):
            w, back = _inplace(ai)
            orc.mont_enter(_np(w), _np(r.contiguous()), w.size(0), _np(l), _np(h), _np(kl_), _np(kh_))
            if back is not None:
                back.copy_(w)

    def mont_redc(a, ql, qh, kl, kh):
        for ai, l, h, kl_, kh_ in zip(a, ql, qh, kl, kh):
            w, back = _inplace(ai)
            orc.mont_redc(_np(w), w.size(0), _np(l), _np(h), _np(kl_), _np(kh_))
            if back is not None:
                back.copy_(w)

    def _fwd(w, e, o, p, rows, q2, l, h, kl_, kh_):
        p = p.contiguous()
        if p.dim() == 2:   # compact [rows, N] table
            logN = w.size(-1).bit_length() - 1
            orc.ntt(_np(w), _np(p), rows, logN, _np(q2), _np(l), _np(h), _np(kl_), _np(kh_))
        else:              # the reference's [rows, logN, N/2] table + gather indices
            orc.ntt_tab(_np(w), _np(e), _np(o), _np(p), rows, _np(q2), _np(l), _np(h), _np(kl_), _np(kh_))

    def ntt(a, even, odd, psi, _2q, ql, qh, kl, kh):
        for ai, e, o, p, q2, l, h, kl_, kh_ in zip(a, even, odd, psi, _2q, ql, qh, kl, kh):
            w, back = _inplace(ai)
            _fwd(w, e, o, p, l.size(0), q2, l, h, kl_, kh_)   # extent = ql.size(0), K.cu:298
            if back is not None:
                back.copy_(w)

    def enter_ntt(a, Rs, even, odd, psi, _2q, ql, qh, kl, kh):
        for ai, r, e, o, p, q2, l, h, kl_, kh_ in zip(a, Rs, even, odd, psi, _2q, ql, qh, kl, kh):
            w, back = _inplace(ai)
            rows = l.size(0)
            orc.mont_enter(_np(w)[:rows], _np(r.contiguous()), rows, _np(l), _np(h), _np(kl_), _np(kh_))
            _fwd(w, e, o, p, rows, q2, l, h, kl_, kh_)
            if back is not None:
                back.copy_(w)

    def _chain(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh, tail):
        for ai, e, o, p, ni, q2, l, h, kl_, kh_ in zip(a, even, odd, psi, Ninv, _2q, ql, qh, kl, kh):
            w, back = _inplace(ai)
            rows = l.size(0)
            p = p.contiguous()
            if p.dim() == 2:
                logN = w.size(-1).bit_length() - 1
                orc.intt(_np(w), _np(p), _np(ni.contiguous()), rows, logN, _np(q2), _np(l), _np(h), _np(kl_), _np(kh_))
            else:
                orc.intt_tab(_np(w), _np(e), _np(o), _np(p), _np(ni.contiguous()), rows, _np(q2), _np(l), _np(h), _np(kl_), _np(kh_))
            view = _np(w)[:rows]   # chain tails run over the transformed rows (K.cu:709-973)
            if tail >= 1:
                orc.mont_redc(view, rows, _np(l), _np(h), _np(kl_), _np(kh_))
            if tail >= 2:
                orc.reduce_2q(view, rows, _np(q2))
            if tail >= 3:
                orc.make_signed(view, rows, _np(q2))
            if back is not None:
                back.copy_(w)

    m.intt = lambda a, *r: _chain(a, *r, tail=0)
    m.intt_exit = lambda a, *r: _chain(a, *r, tail=1)
    m.intt_exit_reduce = lambda a, *r: _chain(a, *r, tail=2)
    m.intt_exit_reduce_signed = lambda a, *r: _chain(a, *r, tail=3)

    def _fix(fn):
        def op(a, _2q):
            for ai, q2 in zip(a, _2q):
                w, back = _inplace(ai)
                fn(_np(w), w.size(0), _np(q2.contiguous()))
                if back is not None:
                    back.copy_(w)
        return op

    def _bin(fn):
        def op(a, b, _2q):
            out = []
            for ai, bi, q2 in zip(a, b, _2q):
                ai_c = ai.contiguous()
                c = torch.empty_like(ai_c)
                fn(_np(ai_c), _np(bi.contiguous()), _np(c), ai_c.size(0), _np(q2.contiguous()))
                out.append(c)
            return out
        return op

    def tile_unsigned(a, _2q):
        out = []
        for ai, q2 in zip(a, _2q):
            ai.squeeze_()
            src = ai.contiguous()
            c = src.new_empty((q2.size(0), src.size(0)))
            orc.tile_unsigned(_np(src), _np(c), q2.size(0), _np(q2.contiguous()))
            out.append(c)
        return out

    m.mont_mult, m.mont_enter, m.mont_redc, m.ntt, m.enter_ntt = mont_mult, mont_enter, mont_redc, ntt, enter_ntt
    m.reduce_2q, m.make_signed, m.make_unsigned = _fix(orc.reduce_2q), _fix(orc.make_signed), _fix(orc.make_unsigned)
    m.mont_add, m.mont_sub = _bin(orc.mont_add), _bin(orc.mont_sub)
    m.tile_unsigned = tile_unsigned
    return m


class OracleBackend:
    """Fused engine ops as sequences of oracle primitives (reference composition, one device)."""
    name = "oracle-cpu"
    host_tensors = True      # this checker keeps "device" tensors in host memory (engine: cpu()/save() heuristics)

    def __init__(self):
        from tests.oracle_csprng import oracle_csprng_class
        self.ops = make_ops()
        self.csprng_class = oracle_csprng_class()     # product host logic over the checker's sampler kernels

    @staticmethod
    def _m(c):
        return _np(c.ql), _np(c.qh), _np(c.kl), _np(c.kh)

    # ---- NTT family ----
    def ntt(self, buf, batch, rows, logN, psi, Rs, c, relaxed=False, plain=False):
        v = _np(buf).reshape(batch, -1, buf.size(-1))
        for b in range(batch):
            x = v[b][:rows]
            if Rs is not None:
                orc.mont_enter(x, _np(Rs), rows, *self._m(c))
            orc.ntt(x, _np(psi), rows, logN, _np(c._2q), *self._m(c))

    def intt(self, buf, batch, rows, logN, ipsi, Ninv, tail, c, relaxed=False, plain=False):
        v = _np(buf).reshape(batch, -1, buf.size(-1))
        for b in range(batch):
            x = v[b][:rows]
            orc.intt(x, _np(ipsi), _np(Ninv), rows, logN, _np(c._2q), *self._m(c))
            if tail >= 1:
                orc.mont_redc(x, rows, *self._m(c))
            if tail >= 2:
                orc.reduce_2q(x, rows, _np(c._2q))
            if tail >= 3:
                orc.make_signed(x, rows, _np(c._2q))

    relin_fold = True

    def intt_mul(self, dst, a, b, batch, rows, logN, ipsi, Ninv, c, a_stride=None, b_stride=None, plain=True):
        """Checker form of lf_intt_mul: mont_mult then intt_exit_reduce, polynomial by polynomial."""
        N = dst.size(-1)
        D = _np(dst).reshape(batch, -1, N)
        fa, fb = a.reshape(-1), b.reshape(-1)
        sa, sb = (rows * N if a_stride is None else a_stride), (rows * N if b_stride is None else b_stride)
        for p in range(batch):
            A = np.ascontiguousarray(_np(torch.as_strided(fa, (rows, N), (N, 1), fa.storage_offset() + p * sa)))
            B = np.ascontiguousarray(_np(torch.as_strided(fb, (rows, N), (N, 1), fb.storage_offset() + p * sb)))
            prod = np.empty_like(A)
            orc.mont_mult(A, B, prod, rows, *self._m(c))
            D[p][:rows] = prod
        self.intt(dst, batch, rows, logN, ipsi, Ninv, 2, c)

    def _fold(self, s, fold, c):
        """s[0] += P * x0 y0, s[1] += P * (x0 y1 + x1 y0) on the ordinary rows, NTT domain, Montgomery form (RelinFold)."""
        x, PR = fold[0], fold[1]
        ell = x.size(1)
        cut = lambda v: np.ascontiguousarray(_np(v)[:ell])
        ql, qh, kl, kh = (cut(v) for v in (c.ql, c.qh, c.kl, c.kh))
        q2 = cut(c._2q)
        x0, x1, y0, y1 = (np.ascontiguousarray(_np(x[i])) for i in range(4))
        d0, t0, t1, d1 = (np.empty_like(x0) for _ in range(4))
        orc.mont_mult(x0, y0, d0, ell, ql, qh, kl, kh)
        orc.mont_mult(x0, y1, t0, ell, ql, qh, kl, kh)
        orc.mont_mult(x1, y0, t1, ell, ql, qh, kl, kh)
        orc.mont_add(t0, t1, d1, ell, q2)
        pr = np.ascontiguousarray(_np(PR)[:ell])
        for comp, dd in ((0, d0), (1, d1)):
            orc.mont_enter(dd, pr, ell, ql, qh, kl, kh)               # REDC(d R * P R) = d P R
            cur = np.ascontiguousarray(_np(s[comp])[:ell])
            nxt = np.empty_like(cur)
            orc.mont_add(cur, dd, nxt, ell, q2)
            _np(s[comp])[:ell] = nxt

    def galois(self, a, dst, rows, logN, p, _2q):
        # encdec.rotate (+ make_unsigned, reduce_2q when _2q is given: ckks_engine.py:1194-1200)
        orc.galois(_np(a)[:rows], _np(dst)[:rows], rows, p)
        if _2q is not None:
            orc.make_unsigned(_np(dst)[:rows], rows, _np(_2q))
            orc.reduce_2q(_np(dst)[:rows], rows, _np(_2q))

    # ---- rescale: ckks_engine.py:1017-1041 ----
    def rescale(self, src, row0, out, rows, scales, round_at, c):
        data = _np(src)[:rows] - _np(row0)[None, :]                  # data - rescaler (torch sub)
        data = np.ascontiguousarray(data)
        orc.mont_enter(data, _np(scales), rows, *self._m(c))         # mont_enter_scalar(rescale_scales)
        data += (_np(row0)[None, :] > round_at).astype(np.int64)     # + rounder
        orc.reduce_2q(data, rows, _np(c._2q))
        _np(out)[:rows] = data

    def rescale_batch(self, srcs, row0s, outs, rows, scales, round_at, c):
        for src, row0, out in zip(srcs, row0s, outs):
            self.rescale(src, row0, out, rows, scales, round_at, c)

    def rescale_ntt(self, srcs, row0s, buf, rows, logN, scales, round_at, psi, Rs, c, relaxed=False, plain=False):
        self.rescale_batch(srcs, row0s, [buf[i] for i in range(len(srcs))], rows, scales, round_at, c)
        self.ntt(buf, len(srcs), rows, logN, psi, Rs, c, relaxed=relaxed, plain=plain)

    def _permuted(self, t, rows, galois):
        """t(X^p) as a new tensor, with the reference's scatter (oracle galois), p from p^-1."""
        pinv, g2q = galois
        N = t.size(-1)
        out = torch.empty_like(t[:rows])
        self.galois(t[:rows].contiguous(), out, rows, N.bit_length() - 1, pow(pinv, -1, 2 * N), g2q)
        return out

    @staticmethod
    def moddown_ws_words(count, ell, K, N):
        return 1

    def ks_moddown_ws(self, ss, outs, addends, ell, K, ws, PiR, Rs, c, PiP=None, galois=None):
        self.ks_moddown_batch(ss, outs, addends, ell, K, PiR, Rs, c, PiP=PiP, galois=galois)

    def ks_moddown_batch(self, ss, outs, addends, ell, K, PiR, Rs, c, PiP=None, galois=None):
        for s_, out, add in zip(ss, outs, addends):
            if add is not None and galois is not None:
                add = self._permuted(add, ell, galois)
            self.ks_moddown(s_, out, add, ell, K, PiR, Rs, c, PiP=PiP)

    def galois_batch(self, srcs, dsts, rows, logN, p, _2q):
        for a, d in zip(srcs, dsts):
            self.galois(a, d, rows, logN, p, _2q)

    # ---- tensor product: ckks_engine.py:1095-1101 ----
    def tensor(self, x0, x1, y0, y1, d0, d1, d2, rows, c, plain=False):
        a0, a1, b0, b1 = (np.ascontiguousarray(_np(t)[:rows]) for t in (x0, x1, y0, y1))
        t0, t1 = np.empty_like(a0), np.empty_like(a0)
        orc.mont_mult(a0, b0, t0, rows, *self._m(c)); _np(d0)[:rows] = t0
        orc.mont_mult(a0, b1, t0, rows, *self._m(c))
        orc.mont_mult(a1, b0, t1, rows, *self._m(c))
        s = np.empty_like(a0)
        orc.mont_add(t0, t1, s, rows, _np(c._2q)); _np(d1)[:rows] = s
        orc.mont_mult(a1, b1, t0, rows, *self._m(c)); _np(d2)[:rows] = t0

    # ---- pre_extend: ckks_engine.py:654-705 ----
    def ks_digits(self, a, state, nparts, desc, tab, c, galois=None):
        if galois is not None:
            a = self._permuted(a, a.size(0), galois)
        A, S = _np(a), _np(state)
        ql, qh, kl, kh = self._m(c)
        D, T = _np(desc).reshape(-1, 4), _np(tab)
        for p in range(nparts):
            r0, alpha, y_off, l_off = (int(x) for x in D[p])
            part = A[r0:r0 + alpha]
            st = np.repeat(part[0:1], alpha, axis=0).copy()          # state = a_part[0].repeat(alpha, 1)
            lc = 0
            for i in range(alpha - 1):
                row = r0 + i + 1
                Y = np.ascontiguousarray((part[i + 1] - st[i + 1])[None, :])
                orc.mont_enter(Y, T[y_off + i:y_off + i + 1].copy(), 1, ql[row:row + 1].copy(), qh[row:row + 1].copy(),
                               kl[row:row + 1].copy(), kh[row:row + 1].copy())
                st[i + 1] = Y[0]
                n_new = alpha - (i + 2)
                if n_new > 0:
                    new = np.repeat(Y, n_new, axis=0).copy()
                    lo = r0 + i + 2
                    orc.mont_enter(new, T[l_off + lc:l_off + lc + n_new].copy(), n_new, ql[lo:lo + n_new].copy(),
                                   qh[lo:lo + n_new].copy(), kl[lo:lo + n_new].copy(), kh[lo:lo + n_new].copy())
                    st[i + 2:] += new
                    lc += n_new
            S[r0:r0 + alpha] = st

    # ---- extend: ckks_engine.py:707-743 ----
    def ks_extend(self, state, ext, nparts, rows, desc, E, c):
        S, X = _np(state), _np(ext)
        D, Et = _np(desc).reshape(-1, 3), _np(E)
        for p in range(nparts):
            r0, alpha, e_off = (int(x) for x in D[p])
            alpha &= 0xff   # bit 8 is a hint for the HIP fused core (digit words wider than 53 bits)
            acc = np.repeat(S[r0:r0 + 1], rows, axis=0).copy()
            orc.mont_enter(acc, Et[e_off:e_off + rows].copy(), rows, *self._m(c))
            for i in range(1, alpha):
                Y = np.repeat(S[r0 + i:r0 + i + 1], rows, axis=0).copy()
                orc.mont_enter(Y, Et[e_off + i * rows:e_off + (i + 1) * rows].copy(), rows, *self._m(c))
                nxt = np.empty_like(acc)
                orc.mont_add(acc, Y, nxt, rows, _np(c._2q))
                acc = nxt
            X[p, :rows] = acc

    # ---- switcher_later_part products + sum over parts: ckks_engine.py:931-934, 832-840 ----
    def ks_inner(self, ext, key, first_part, row_off, s0, s1, nparts, rows, c):
        X, Kt = _np(ext), _np(key)
        for comp, dst in ((0, s0), (1, s1)):
            acc = None
            for p in range(nparts):
                e = np.ascontiguousarray(X[p, :rows])
                k = np.ascontiguousarray(Kt[first_part + p, comp, row_off:row_off + rows])
                prod = np.empty_like(e)
                orc.mont_mult(e, k, prod, rows, *self._m(c))
                if acc is None:
                    acc = prod
                else:
                    nxt = np.empty_like(acc)
                    orc.mont_add(acc, prod, nxt, rows, _np(c._2q))
                    acc = nxt
            _np(dst)[:rows] = acc

    fused_ks_min_logN = 13

    def ks_core(self, state, nparts, rows, logN, desc, E, Ed, key, first_part, row_off, tmp, s, psi, ipsi, Ninv, c, fold=None):
        """The fused core as the sequence it replaces: extend, NTT, inner product + sum, inverse NTT chain."""
        self.ks_extend(state, tmp, nparts, rows, desc, E, c)
        self.ntt(tmp, nparts, rows, logN, psi, None, c)
        self.ks_inner(tmp, key, first_part, row_off, s[0], s[1], nparts, rows, c)
        if fold is not None:
            self._fold(s, fold, c)
        self.intt(s, 2, rows, logN, ipsi, Ninv, 2, c)

    def ks_fwd(self, state, first, count, rows, logN, desc, E, Ed, tmp, psi, c, own=None):
        """Digits first .. first + count - 1 only: extend + NTT into tmp[first:first + count]."""
        D = desc.reshape(-1, 3)[first:first + count].contiguous()
        sub = tmp[first:first + count]
        self.ks_extend(state, sub, count, rows, D, E, c)
        self.ntt(sub, count, rows, logN, psi, None, c)

    def ks_tail(self, nparts, rows, logN, key, first_part, row_off, tmp, s, ipsi, Ninv, c, fold=None):
        self.ks_inner(tmp, key, first_part, row_off, s[0], s[1], nparts, rows, c)
        if fold is not None:
            self._fold(s, fold, c)
        self.intt(s, 2, rows, logN, ipsi, Ninv, 2, c)

    ks_batch_sizes = (4, 2)

    def ks_digits_batch(self, srcs, states, nparts, desc, tab, c, galois=None):
        for src, st in zip(srcs, states):
            self.ks_digits(src, st, nparts, desc, tab, c, galois=galois)

    def ks_core_batch(self, states, nparts, rows, logN, desc, E, Ed, key, first_part, row_off, tmp, s, psi, ipsi, Ninv, c, fold=None):
        for b in range(states.size(0)):
            f = None if fold is None else (fold[0][b], fold[1], fold[2])
            self.ks_core(states[b], nparts, rows, logN, desc, E, Ed, key, first_part, row_off, tmp[b], s[b], psi, ipsi, Ninv, c,
                         fold=f)

    # ---- divide by P: ckks_engine.py:850-901 (+ relinearize 1135-1140 / switch_key 952-953) ----
    def ks_moddown(self, s, out, addend, ell, K, PiR, Rs, c, PiP=None):
        rows = ell + K
        d = np.ascontiguousarray(_np(s)[:rows]).copy()
        ql, qh, kl, kh = self._m(c)
        q2, rs, pir = _np(c._2q), _np(Rs), _np(PiR)
        cut = lambda v, n: np.ascontiguousarray(v[:n])
        orc.mont_enter(d[:ell], cut(rs, ell), ell, cut(ql, ell), cut(qh, ell), cut(kl, ell), cut(kh, ell))
        for P_ind in range(K):
            P = np.repeat(d[rows - 1 - P_ind:rows - P_ind], rows, axis=0).copy()
            orc.mont_enter(P[:ell], cut(rs, ell), ell, cut(ql, ell), cut(qh, ell), cut(kl, ell), cut(kh, ell))
            nxt = np.empty_like(d)
            orc.mont_sub(d, P, nxt, rows, q2)
            d = nxt
            orc.mont_enter(d, np.ascontiguousarray(pir[P_ind]), rows, ql, qh, kl, kh)
            orc.reduce_2q(d, rows, q2)
        c_ = np.ascontiguousarray(d[:ell])
        orc.mont_redc(c_, ell, cut(ql, ell), cut(qh, ell), cut(kl, ell), cut(kh, ell))
        orc.reduce_2q(c_, ell, cut(q2, ell))
        if addend is not None:
            c_ = c_ + _np(addend)[:ell]
            orc.reduce_2q(c_, ell, cut(q2, ell))
        _np(out)[:ell] = c_
