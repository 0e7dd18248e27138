"""Device arithmetic behind the engine: thin tensor-level wrappers over the C ABI (include/ckks_hip.h).

`HipBackend` is the only backend in the package and it has no fallback — every method launches a HIP
kernel through libckks_hip.so on the tensor's GPU and current torch stream.  The engine takes the
backend as an object so that the multi-rank orchestration (which is backend-agnostic Python) can be
exercised by the CPU test-suite with a checker backend that lives under tests/.
"""
from __future__ import annotations

import ctypes

import torch

from .._native import lib, check, KsPlan, LF_KEY_PLANES
from ..ntt import ntt_cuda, twiddles


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the stream handle without a Stream object per call


def _ds(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError(f"HipBackend: tensor on {t.device}; device memory required (no CPU fallback)")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if _raw_stream is not None:
        return idx, _raw_stream(idx)
    return idx, torch.cuda.current_stream(idx).cuda_stream


def _p(t):
    if t is None:
        return 0
    if t.dtype != torch.int64:
        raise TypeError(f"HipBackend: int64 tensor required, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("HipBackend: contiguous tensor required (the kernels take a raw pointer and a row pitch of N words)")
    return t.data_ptr()


def _pb(t):
    """Device pointer of a uint8 table (None -> NULL)."""
    if t is None:
        return 0
    if t.dtype != torch.uint8 or not t.is_contiguous():
        raise TypeError("HipBackend: contiguous uint8 tensor required")
    return t.data_ptr()


def _pd(t):
    """Device pointer of a float64 table (None -> NULL)."""
    if t is None:
        return 0
    if t.dtype != torch.float64:
        raise TypeError(f"HipBackend: float64 tensor required, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("HipBackend: contiguous tensor required")
    return t.data_ptr()


def _parr(tensors):
    """Host array of device pointers (None -> NULL) for the batched entry points."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else _p(t)
    return arr


class Consts:
    """Per-row Montgomery constants of a contiguous run of limbs on one device."""
    __slots__ = ("ql", "qh", "kl", "kh", "_2q", "q_host", "_mont", "_qptr")

    def __init__(self, ql, qh, kl, kh, _2q, q_host=None):
        self.ql, self.qh, self.kl, self.kh, self._2q = ql, qh, kl, kh, _2q
        self.q_host = q_host   # numpy int64 copy of the primes (launch-time row classification)
        # the vectors live as long as the object: validated and resolved once, not on every launch
        self._mont = (_p(ql), _p(qh), _p(kl), _p(kh))
        self._qptr = 0 if q_host is None else q_host.ctypes.data

    def qptr(self, relaxed=False):
        if relaxed and not self._qptr:
            raise ValueError("relaxed transforms need the host primes (Consts(q_host=...)): the auxiliary twiddle rows "
                             "are laid out by the size of the prime")
        return self._qptr

    def mont(self):
        return self._mont


class HipBackend:
    name = "hip-gfx950"
    ops = ntt_cuda  # the 15 reference-shaped primitives
    # capacities compiled into the fused kernels (include/ckks_hip.h: lf_limits)
    limits = {"digit_limbs": int(lib.lf_limits(0)), "special_primes": int(lib.lf_limits(1)), "rows": int(lib.lf_limits(2)),
              "batch": int(lib.lf_limits(3)), "logN": int(lib.lf_limits(4))}

    def warm_twiddles(self, table, c: "Consts"):
        """Build the fp64 twin of a twiddle table on the current stream (otherwise built by its first user)."""
        dev, st = _ds(table)
        twiddles.dp_pointer(table, c.ql, c.qh, c.kl, c.kh, dev, st)

    # ---- NTT family over [batch][rows][N] stacks ------------------------------------------------
    def ntt(self, buf, batch, rows, logN, psi, Rs, c: Consts, relaxed=False, plain=False):
        """relaxed: the caller only needs residues mod q (internal transforms whose consumers reduce).
        plain (with relaxed): fp64-class limbs stay in the plain domain (see LF_NTT_PLAIN)."""
        dev, st = _ds(buf)
        dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        check(lib.lf_ntt(_p(buf), batch, rows, logN, _p(psi), dp, c.qptr(relaxed), _p(Rs), (1 if relaxed else 0) | (2 if relaxed and plain else 0), _p(c._2q),
                         *c.mont(), dev, st), "lf_ntt")

    def rescale_ntt(self, srcs, row0s, buf, rows, logN, scales, round_at, psi, Rs, c: Consts, relaxed=False, plain=False):
        """Rescale len(srcs) polynomials into buf[i] and transform them (cc_mult's opening) in one call."""
        dev, st = _ds(buf)
        dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        check(lib.lf_rescale_ntt(_parr(srcs), _parr(row0s), len(srcs), _p(buf), rows, logN, _p(scales), round_at, _p(psi), dp,
                                 c.qptr(relaxed), _p(Rs), (1 if relaxed else 0) | (2 if relaxed and plain else 0), _p(c._2q),
                                 *c.mont(), dev, st), "lf_rescale_ntt")

    def intt(self, buf, batch, rows, logN, ipsi, Ninv, tail, c: Consts, relaxed=False, plain=False):
        dev, st = _ds(buf)
        dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        check(lib.lf_intt(_p(buf), batch, rows, logN, _p(ipsi), dp, c.qptr(relaxed and tail >= 2), _p(Ninv), tail,
                          (3 if plain else 1) if relaxed and tail >= 2 else 0, _p(c._2q), *c.mont(), dev, st), "lf_intt")

    def intt_mul(self, dst, a, b, batch, rows, logN, ipsi, Ninv, c: Consts, a_stride=None, b_stride=None, plain=True):
        """dst[p] = intt(a[p] * b[p]) to canonical coefficients (relaxed, tail 2): the product is formed as the first pass
        loads its tiles.  a, b: first polynomial of each factor; *_stride = words between a factor's polynomials."""
        dev, st = _ds(dst)
        N = dst.size(-1)
        dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        check(lib.lf_intt_mul(_p(dst), _p(a), rows * N if a_stride is None else a_stride, _p(b),
                              rows * N if b_stride is None else b_stride, batch, rows, logN, _p(ipsi), dp, c.qptr(True), _p(Ninv), 2,
                              3 if plain else 1, *c.mont(), dev, st), "lf_intt_mul")

    def galois(self, a, dst, rows, logN, p, _2q):
        dev, st = _ds(a)
        check(lib.lf_galois(_p(a), _p(dst), rows, logN, p, _p(_2q), dev, st), "lf_galois")

    def galois_batch(self, srcs, dsts, rows, logN, p, _2q):
        """Several polynomials (the components of a ciphertext) through the same permutation in one launch."""
        dev, st = _ds(dsts[0])
        check(lib.lf_galois_batch(_parr(srcs), _parr(dsts), len(srcs), rows, logN, p, _p(_2q), dev, st), "lf_galois_batch")

    def gather_rows(self, rows, dst):
        """len(rows) <= 8 row tensors [N] -> the consecutive rows of dst [len(rows), N]: one launch."""
        dev, st = _ds(dst)
        check(lib.lf_gather_rows(_parr(rows), _p(dst), len(rows), dst.size(-1), dev, st), "lf_gather_rows")

    # ---- fused engine ops ------------------------------------------------------------------------
    def rescale_batch(self, srcs, row0s, outs, rows, scales, round_at, c: Consts):
        dev, st = _ds(outs[0])
        check(lib.lf_rescale_batch(_parr(srcs), _parr(row0s), _parr(outs), len(srcs), rows, outs[0].size(-1), _p(scales),
                                   round_at, *c.mont(), dev, st), "lf_rescale_batch")

    def ks_moddown_batch(self, ss, outs, addends, ell, K, PiR, Rs, c: Consts, PiP=None, galois=None):
        """galois = (p^-1 mod 2N, _2q or None): the addends are added as addend(X^p), read in gather form."""
        dev, st = _ds(outs[0])
        pinv, g2q = (0, None) if galois is None else galois
        check(lib.lf_ks_moddown_batch(_parr(ss), _parr(outs), _parr(addends), len(ss), ell, K, outs[0].size(-1), _p(PiR),
                                      _pd(PiP), _p(Rs), pinv, _p(g2q), *c.mont(), dev, st),
              "lf_ks_moddown_batch")

    def ks_moddown_ws(self, ss, outs, addends, ell, K, ws, PiR, Rs, c: Consts, PiP=None, galois=None, one_launch=False):
        """ks_moddown_batch with a workspace tensor `ws` (int64, >= moddown_ws_words(..) words): the special-prime
        chain is evaluated once per coefficient by a first launch — or, one_launch (K <= moddown_one_max_K, constants
        already in `ws` through moddown_consts): inside the single mod-down launch."""
        dev, st = _ds(outs[0])
        pinv, g2q = (0, None) if galois is None else galois
        fn, what = (lib.lf_ks_moddown_one, "lf_ks_moddown_one") if one_launch else (lib.lf_ks_moddown_ws, "lf_ks_moddown_ws")
        check(fn(_parr(ss), _parr(outs), _parr(addends), len(ss), ell, K, outs[0].size(-1), _p(ws), ws.numel(), _p(PiR), _pd(PiP),
                 _p(Rs), pinv, _p(g2q), *c.mont(), dev, st), what)

    moddown_one_max_K = 2    # LF_MODDOWN_ONE_MAX_K: up to here the mod-down is ONE launch (pivots eliminated inside it)

    def moddown_consts(self, ws, count, ell, K, N, PiP, c: Consts):
        """Write the level constants of the mod-down's fp64 rows into workspace `ws`, ONCE per workspace (needed by
        ks_moddown_ws(.., one_launch=True); the two-launch form rewrites them on every call)."""
        dev, st = _ds(ws)
        check(lib.lf_ks_moddown_consts(_p(ws), ws.numel(), count, ell, K, N, _pd(PiP), *c.mont(), dev, st), "lf_ks_moddown_consts")

    @staticmethod
    def moddown_ws_words(count, ell, K, N):
        return int(lib.lf_ks_moddown_ws_words(count, ell, K, N))

    def rescale(self, src, row0, out, rows, scales, round_at, c: Consts):
        dev, st = _ds(out)
        check(lib.lf_rescale(_p(src), _p(row0), _p(out), rows, out.size(-1), _p(scales), round_at, *c.mont(), dev, st),
              "lf_rescale")

    def tensor(self, x0, x1, y0, y1, d0, d1, d2, rows, c: Consts, plain=False):
        dev, st = _ds(d0)
        check(lib.lf_tensor(_p(x0), _p(x1), _p(y0), _p(y1), _p(d0), _p(d1), _p(d2), rows, d0.size(-1), 1 if plain else 0,
                            *c.mont(), dev, st),
              "lf_tensor")

    def ks_digits(self, a, state, nparts, desc, tab, c: Consts, galois=None):
        """galois = (p^-1 mod 2N, _2q or None): the digits of a(X^p), read in gather form (no permutation pass)."""
        dev, st = _ds(a)
        if galois is None:
            check(lib.lf_ks_digits(_p(a), _p(state), nparts, _p(desc), _p(tab), a.size(-1), *c.mont(), dev, st), "lf_ks_digits")
        else:
            check(lib.lf_ks_digits_galois(_p(a), _p(state), nparts, _p(desc), _p(tab), a.size(-1), galois[0], _p(galois[1]),
                                          *c.mont(), dev, st), "lf_ks_digits_galois")

    def ks_digits_batch(self, srcs, states, nparts, desc, tab, c: Consts, galois=None):
        """ks_digits of len(srcs) (<= 8) polynomials in one launch."""
        dev, st = _ds(srcs[0])
        pinv, g2q = (0, None) if galois is None else galois
        check(lib.lf_ks_digits_batch(_parr(srcs), _parr(states), len(srcs), nparts, _p(desc), _p(tab), srcs[0].size(-1), pinv,
                                     _p(g2q), *c.mont(), dev, st), "lf_ks_digits_batch")

    def ks_extend(self, state, ext, nparts, rows, desc, E, c: Consts):
        dev, st = _ds(ext)
        check(lib.lf_ks_extend(_p(state), _p(ext), nparts, rows, ext.size(-1), _p(desc), _p(E), *c.mont(), dev, st),
              "lf_ks_extend")

    def ks_inner(self, ext, key, first_part, row_off, s0, s1, nparts, rows, c: Consts):
        """key: packed [parts, 2, rows0, N] tensor of this device."""
        dev, st = _ds(ext)
        N = ext.size(-1)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        check(lib.lf_ks_inner(_p(ext), base, part_stride, comp_stride, row_off, _p(s0), _p(s1), nparts, rows, N,
                              *c.mont(), dev, st), "lf_ks_inner")

    @staticmethod
    def _kfmt(key):
        """Format of a packed key tensor (LF_KEY_RAW unless key_planes() made it)."""
        return getattr(key, "lf_key_format", 0)

    def key_planes(self, src_b, src_a, dst_b, dst_a, c: Consts):
        """The two [rows, N] components of one key part (any lazy words) -> the planes format the fused key-switch entries
        read with key_format = LF_KEY_PLANES (include/ckks_hip.h): fp64-class rows as interleaved 32-bit + 16-bit planes of
        the canonical residues of both components, integer-class rows raw."""
        dev, st = _ds(dst_b)
        check(lib.lf_key_planes(_p(src_b), _p(src_a), _p(dst_b), _p(dst_a), src_b.size(0), src_b.size(1), _p(c.ql), _p(c.qh), dev, st),
              "lf_key_planes")

    @staticmethod
    def mark_planes(t):
        t.lf_key_format = LF_KEY_PLANES
        return t

    fused_ks_min_logN = 13   # lf_ks_core needs a two-pass ring degree
    relin_fold = True        # cc_mult's d0 / d1 folded into the key-switch sums (lf_intt_mul, lf_relin_*)

    def ks_core(self, state, nparts, rows, logN, desc, E, Ed, key, first_part, row_off, tmp, s, psi, ipsi, Ninv,
                c: Consts, fold=None):
        """extend + NTT + key inner product + inverse NTT in three fused launches per arithmetic class.
        fold = (x stack [4, ell, N], PR [ell], own [rows] uint8 or None): cc_mult's d0 / d1 enter the sums in the NTT domain and
        the digits' own limbs come from x1 * y1 instead of an extension (lf_relin_core_batch)."""
        dev, st = _ds(s)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        psi_dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        ipsi_dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        if fold is not None:
            x, PR, own = fold
            check(lib.lf_relin_core_batch(_p(state), 0, 1, nparts, rows, logN, _p(desc), _p(E), _pd(Ed), base, part_stride,
                                          comp_stride, row_off, self._kfmt(key), _p(tmp), _p(s), _p(psi), psi_dp, _p(ipsi), ipsi_dp, _p(Ninv),
                                          _p(x), 0, _p(PR), x.size(1), _pb(own), c.qptr(), *c.mont(), dev, st),
                  "lf_relin_core_batch")
            return
        check(lib.lf_ks_core(_p(state), nparts, rows, logN, _p(desc), _p(E), _pd(Ed), base, part_stride, comp_stride,
                             row_off, self._kfmt(key), _p(tmp), _p(s), _p(psi), psi_dp, _p(ipsi), ipsi_dp, _p(Ninv),
                             c.qptr(), *c.mont(), dev, st), "lf_ks_core")

    def ks_fwd(self, state, first, count, rows, logN, desc, E, Ed, tmp, psi, c: Consts, own=None):
        """Extension + forward NTT of digits first .. first + count - 1 into tmp[first:first + count] (lf_ks_fwd; with
        `own`, lf_relin_fwd: the digits' own limbs are left out, see ks_core's fold)."""
        dev, st = _ds(tmp)
        psi_dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        N = tmp.size(-1)
        if own is not None:
            check(lib.lf_relin_fwd(_p(state), first, count, rows, logN, _p(desc), _p(E), _pd(Ed), _p(tmp), _p(psi), psi_dp,
                                   _pb(own), c.qptr(), *c.mont(), dev, st), "lf_relin_fwd")
            return
        check(lib.lf_ks_fwd(_p(state), count, rows, logN, _p(desc) + first * 3 * 8, _p(E), _pd(Ed),
                            _p(tmp) + first * rows * N * 8, _p(psi), psi_dp, c.qptr(), *c.mont(), dev, st), "lf_ks_fwd")

    def ks_tail(self, nparts, rows, logN, key, first_part, row_off, tmp, s, ipsi, Ninv, c: Consts, fold=None):
        """Inner product of all extended digits with the key + inverse NTT (lf_ks_tail; fold: lf_relin_tail, see ks_core)."""
        dev, st = _ds(s)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        ipsi_dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        if fold is not None:
            x, PR, own = fold
            check(lib.lf_relin_tail(nparts, rows, logN, base, part_stride, comp_stride, row_off, self._kfmt(key), _p(tmp), _p(s), _p(ipsi), ipsi_dp,
                                    _p(Ninv), _p(x), _p(PR), x.size(1), _pb(own), c.qptr(), *c.mont(), dev, st), "lf_relin_tail")
            return
        check(lib.lf_ks_tail(nparts, rows, logN, base, part_stride, comp_stride, row_off, self._kfmt(key), _p(tmp), _p(s), _p(ipsi), ipsi_dp,
                             _p(Ninv), c.qptr(), *c.mont(), dev, st), "lf_ks_tail")

    # ---- whole ops behind one native call (lf_cc_mult_evk / lf_switch_key over an lf_ks_plan) ------------------------
    native_ops = True

    def make_plan(self, ints, tensors, q_host, psi, ipsi, c: Consts):
        """lf_ks_plan from the engine's per-level pieces.  ints: {field: int}; tensors: {field: tensor or None} (int64 /
        float64 / uint8 device tensors); q_host: numpy int64; psi / ipsi: the twiddle views (their auxiliary tables are
        built here, on the current stream).  Returns (plan, keep-alive list)."""
        plan = KsPlan()
        for k, v in ints.items():
            setattr(plan, k, int(v))
        keep = [q_host, psi, ipsi, c]
        for k, t in tensors.items():
            if t is None:
                setattr(plan, k, None)
                continue
            if not t.is_contiguous():
                raise ValueError(f"plan tensor {k} must be contiguous")
            setattr(plan, k, t.data_ptr())
            keep.append(t)
        dev, st = _ds(psi)
        plan.device = dev
        plan.ql, plan.qh, plan.kl, plan.kh = c.mont()
        plan._2q = _p(c._2q)
        plan.q_host = q_host.ctypes.data
        plan.psi, plan.ipsi = _p(psi), _p(ipsi)
        plan.psi_dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        plan.ipsi_dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        # the plan outlives this call (the engine caches it per level): it holds the auxiliary twins its two raw
        # addresses point into, so a rebuilt twin (new table version) can never leave the plan reading freed memory
        keep += [twiddles.twin_of(psi), twiddles.twin_of(ipsi)]
        if plan.K <= self.moddown_one_max_K:   # the ops then run the one-launch mod-down: its level constants, once
            check(lib.lf_ks_moddown_consts(plan.md_ws, plan.md_ws_words, 2 * plan.max_nct, plan.ell, plan.K, 1 << plan.logN, plan.PiP,
                                           *c.mont(), dev, st), "lf_ks_moddown_consts")
            plan.md_consts = 2 * plan.max_nct   # the op entries take the one-launch form for exactly this count
        return plan, keep

    @staticmethod
    def _key_args(key, first_part):
        part_stride, comp_stride = key.stride(0), key.stride(1)
        return key.data_ptr() + first_part * part_stride * 8, part_stride, comp_stride

    def switch_key_batch_native(self, plan, c0s, c1s, pinv, canonical, key, first_part, row_off, out):
        """len(c1s) in (1, 2, 4) == plan.max_nct ciphertexts under one key as ONE native call; out [nct, 2, ell, N]."""
        dev, st = _ds(out)
        base, ps, cs = self._key_args(key, first_part)
        nct = len(c1s)
        check(lib.lf_switch_key_batch(ctypes.byref(plan), nct, _parr(c0s), _parr(c1s), pinv, 1 if canonical else 0, base, ps, cs, row_off,
                                      self._kfmt(key), _parr([out[t][0] for t in range(nct)]), _parr([out[t][1] for t in range(nct)]), st),
              "lf_switch_key_batch")

    def cc_mult_evk_batch_native(self, plan, ins, row0s, key, first_part, row_off, out):
        """ins / row0s: 4 device tensors (or raw pointers) per ciphertext pair; out [nct, 2, ell, N]."""
        dev, st = _ds(out)
        base, ps, cs = self._key_args(key, first_part)
        nct = out.size(0)
        check(lib.lf_cc_mult_evk_batch(ctypes.byref(plan), nct, ins, row0s, base, ps, cs, row_off, self._kfmt(key),
                                       _parr([out[t][0] for t in range(nct)]), _parr([out[t][1] for t in range(nct)]), st),
              "lf_cc_mult_evk_batch")

    # ---- the halves of an op around the digit exchange (one process per GPU) -----------------------------------------
    def cc_mult_pre(self, plan, ins, row0s, st, which=3):
        """which: 1 = the launch that reads the operands, 2 = the rest of the half (plan addresses only), 3 = both."""
        check(lib.lf_cc_mult_evk_pre(ctypes.byref(plan), ins, row0s, which, st), "lf_cc_mult_evk_pre")

    def switch_key_pre(self, plan, c1, pinv, canonical):
        dev, st = _ds(c1)
        check(lib.lf_switch_key_pre(ctypes.byref(plan), _p(c1), pinv, 1 if canonical else 0, st), "lf_switch_key_pre")

    def plan_fwd(self, plan, digits, first, count, relin):
        dev, st = _ds(digits)
        check(lib.lf_ks_plan_fwd(ctypes.byref(plan), _p(digits), first, count, 1 if relin else 0, st), "lf_ks_plan_fwd")

    def cc_mult_post(self, plan, key, first_part, row_off, out, which=3):
        """which: 1 = inner product + inverse NTT (plan addresses + the key), 2 = the mod-down into `out`, 3 = both
        (out may be None with which = 1)."""
        dev, st = _ds(key)
        base, ps, cs = self._key_args(key, first_part)
        o0 = out.data_ptr() if out is not None else None
        o1 = out.data_ptr() + out.stride(0) * 8 if out is not None else None
        check(lib.lf_cc_mult_evk_post(ctypes.byref(plan), base, ps, cs, row_off, self._kfmt(key), o0, o1, which, st), "lf_cc_mult_evk_post")

    def switch_key_post(self, plan, c0, pinv, canonical, key, first_part, row_off, out, which=3):
        dev, st = _ds(key)
        base, ps, cs = self._key_args(key, first_part)
        o0 = out.data_ptr() if out is not None else None
        o1 = out.data_ptr() + out.stride(0) * 8 if out is not None else None
        check(lib.lf_switch_key_post(ctypes.byref(plan), _p(c0) if c0 is not None else None, pinv, 1 if canonical else 0, base, ps, cs,
                                     row_off, self._kfmt(key), o0, o1, which, st), "lf_switch_key_post")

    def cc_mult_evk(self, plan, ins, row0s, key, first_part, row_off, out):
        """ins / row0s: ctypes arrays of 4 device pointers; out [2, ell, N]."""
        dev, st = _ds(out)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        plane = out.stride(0) * 8
        check(lib.lf_cc_mult_evk(ctypes.byref(plan), ins, row0s, base, part_stride, comp_stride, row_off, self._kfmt(key), out.data_ptr(),
                                 out.data_ptr() + plane, st), "lf_cc_mult_evk")

    def switch_key_native(self, plan, c0, c1, pinv, canonical, key, first_part, row_off, out):
        dev, st = _ds(out)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        plane = out.stride(0) * 8
        check(lib.lf_switch_key(ctypes.byref(plan), _p(c0), _p(c1), pinv, 1 if canonical else 0, base, part_stride, comp_stride,
                                row_off, self._kfmt(key), out.data_ptr(), out.data_ptr() + plane, st), "lf_switch_key")

    ks_batch_sizes = (4, 2)   # ciphertexts per lf_ks_core_batch call (largest first)

    def ks_core_batch(self, states, nparts, rows, logN, desc, E, Ed, key, first_part, row_off, tmp, s, psi, ipsi, Ninv,
                      c: Consts, fold=None):
        """ks_core for len(states) in (2, 4) ciphertexts under one key: states = [nct, state_rows, N] tensor,
        tmp [nct, nparts, rows, N], s [nct, 2, rows, N].  fold = (x [nct, 4, ell, N], PR [ell]), see ks_core."""
        dev, st = _ds(s)
        part_stride, comp_stride = key.stride(0), key.stride(1)
        base = key.data_ptr() + first_part * part_stride * 8
        psi_dp = twiddles.dp_pointer(psi, c.ql, c.qh, c.kl, c.kh, dev, st)
        ipsi_dp = twiddles.dp_pointer(ipsi, c.ql, c.qh, c.kl, c.kh, dev, st)
        if fold is not None:
            x, PR, own = fold
            check(lib.lf_relin_core_batch(_p(states), states.stride(0), states.size(0), nparts, rows, logN, _p(desc), _p(E),
                                          _pd(Ed), base, part_stride, comp_stride, row_off, self._kfmt(key), _p(tmp), _p(s), _p(psi), psi_dp,
                                          _p(ipsi), ipsi_dp, _p(Ninv), _p(x), x.stride(0), _p(PR), x.size(2), _pb(own), c.qptr(),
                                          *c.mont(), dev, st), "lf_relin_core_batch")
            return
        check(lib.lf_ks_core_batch(_p(states), states.stride(0), states.size(0), nparts, rows, logN, _p(desc), _p(E),
                                   _pd(Ed), base, part_stride, comp_stride, row_off, self._kfmt(key), _p(tmp), _p(s), _p(psi), psi_dp,
                                   _p(ipsi), ipsi_dp, _p(Ninv), c.qptr(), *c.mont(), dev, st), "lf_ks_core_batch")

    def ks_moddown(self, s, out, addend, ell, K, PiR, Rs, c: Consts, PiP=None):
        dev, st = _ds(out)
        check(lib.lf_ks_moddown(_p(s), _p(out), _p(addend), ell, K, out.size(-1), _p(PiR),
                                _pd(PiP), _p(Rs), *c.mont(), dev, st), "lf_ks_moddown")
