"""BASELINE configs at their FULL sizes on one MI355X:

  C2  silver, [19, 32768]: enter_ntt then intt_exit_reduce over the whole prime chain — intermediate and final
      words bit-exact vs the oracle, final == input;
      (the same at gold, [39, 65536])
  C5  gold, 64 level-0 ciphertexts rotated under one rotation key in one rotate_single_batch call — every result
      equal to the per-ciphertext rotate_single (which the golden digests pin to the reference engine), and the
      ciphertext the gold fixture covers is part of the batch and checked against its reference digest.
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from liberate_fhe_amd.utils import synth

warnings.filterwarnings("ignore", category=UserWarning)
pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "engine_digests.json")))


@pytest.mark.parametrize("logN,K,rows", [(15, 2, 19), (16, 4, 39)])
def test_full_chain_ntt_round_trip_vs_oracle(logN, K, rows):
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context
    from oracle import oracle as orc
    ctx = ckks_context(logN=logN, num_special_primes=K)
    assert len(ctx.q) == rows
    ntt = ntt_context(ctx, devices=["cuda:0"])
    x = synth.uniform_rows(2, range(rows), ctx.q, ctx.N)            # row i uniform in [0, q_i), seed 2
    dev = torch.from_numpy(x).cuda()
    ntt.enter_ntt([dev], 0, -2)
    got_fwd = dev.cpu().numpy()

    h = lambda v: np.asarray(v, dtype=np.int64)
    ql, qh, kl, kh = h(ctx.q_lower_bits), h(ctx.q_higher_bits), h(ctx.k_lower_bits), h(ctx.k_higher_bits)
    q2, Rs = h(ctx.q_double), h(ctx.R_square)
    psi = np.ascontiguousarray(ctx.psi_br.copy())
    orc.mont_enter(psi, Rs, rows, ql, qh, kl, kh)
    want = x.copy()
    orc.mont_enter(want, Rs, rows, ql, qh, kl, kh)
    orc.ntt(want, psi, rows, ctx.logN, q2, ql, qh, kl, kh)
    assert (got_fwd == want).all(), "enter_ntt: intermediate words differ from the oracle"

    ntt.intt_exit_reduce([dev], 0, -2)
    assert (dev.cpu().numpy() == x).all(), "intt_exit_reduce(enter_ntt(x)) != x"


def _digest(ct):
    from tests.test_engine_golden import digest
    return digest(ct)


def test_gold_rotate_batch_of_64_equals_loop_and_reference_digest():
    from liberate_fhe_amd.fhe import ckks_engine
    rec = GOLD["gold"]
    eng = ckks_engine(devices=["cuda:0"], **rec["params"])
    s = rec["seeds"]
    rotk = synth.key_switch_key(eng, s["rotk"], origin=f"rotation key:{s['rot_delta']}")
    cts = [synth.ciphertext(eng, s["ct_a"], 0)] + [synth.ciphertext(eng, 100 + i, 0) for i in range(63)]
    batch = eng.rotate_single_batch(cts, rotk)
    assert len(batch) == 64
    assert _digest(batch[0]) == rec["ops"]["rotate_single(a,rotk)"]          # the reference engine's output
    for i, ct in enumerate(cts):
        one = eng.rotate_single(ct, rotk)
        for comp in range(2):
            assert torch.equal(one.data[comp][0], batch[i].data[comp][0]), (i, comp)
        del one


@pytest.mark.parametrize("LOGN,L,B,ctx_kw", [(16, 30, 70, dict(num_special_primes=4)),
                                              (13, 28, 600, dict(num_scales=23, num_special_primes=4, is_secured=False))])
def test_large_batch_transform_takes_the_eight_tile_kernel_and_equals_small_batches(LOGN, L, B, ctx_kw):
    """lf_ntt on B polynomials x L limbs — 33 600 tiles both times: above NTT16_SEQ_MIN_BLOCKS x 8, so the tiled pass of
    the EXACT transform runs as ntt_pass16_fwd_seq (8 tiles per block, last-stage twiddles kept in registers) — against the
    same polynomials in two calls of B / 2 (one tile per block); relaxed transforms likewise.  logN 16: tile pairs per class
    are multiples of 8 (XCD-aware block order); logN 13 with 23 + 5 limbs: they are not (plain order).  A few tiles hold
    signed-lazy words (odd-tile path inside a block's loop); polynomial 0 against the oracle."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context, twiddles
    from oracle import oracle as orc
    ctx = ckks_context(logN=LOGN, **ctx_kw)
    ntt = ntt_context(ctx, devices=["cuda:0"])
    total, N = len(ctx.q), ctx.N
    assert total >= L and B * L * (N >> 12) >= 8 * 4096
    rows = list(range(total - L, total))
    host = np.stack([synth.uniform_rows(300 + b, rows, ctx.q, N, lazy=True) for b in range(B)])
    q = np.array([ctx.q[i] for i in rows], dtype=np.int64)
    for b, r, j in ((0, 3, 5), (0, L - 1, N // 2 + 11), (B // 2 + 6, 0, N - 1), (B - 1, 17, N - 4096)):   # signed-lazy words (D.4)
        host[b, r, j] -= 2 * q[r]
    sl = lambda t: t[0][total - L:]
    psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
    st = torch.cuda.current_stream().cuda_stream
    psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)

    def run(x, batch, flags):
        check(lib.lf_ntt(x.data_ptr(), batch, L, LOGN, psi.data_ptr(), psi_dp, q.ctypes.data, 0, flags, q2.data_ptr(),
                         ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st), "lf_ntt")

    for flags in (0, 1):                                   # exact, LF_NTT_RELAXED
        big = torch.from_numpy(host).cuda()
        run(big, B, flags)
        halves = torch.from_numpy(host).cuda()
        run(halves[:B // 2], B // 2, flags)
        run(halves[B // 2:], B - B // 2, flags)
        torch.cuda.synchronize()
        assert torch.equal(big, halves), f"flags {flags}: {B}-polynomial call differs from two half-batches"
        if flags == 0:
            h = lambda v: np.asarray(v, dtype=np.int64)
            pick = lambda v: h(v)[total - L:]
            cs = (pick(ctx.q_lower_bits), pick(ctx.q_higher_bits), pick(ctx.k_lower_bits), pick(ctx.k_higher_bits))
            psi_h = np.ascontiguousarray(ctx.psi_br[total - L:].copy())
            orc.mont_enter(psi_h, pick(ctx.R_square), L, *cs)
            want = host[0].copy()
            orc.ntt(want, psi_h, L, LOGN, pick(ctx.q_double), *cs)
            assert (big[0].cpu().numpy() == want).all(), "polynomial 0 differs from the oracle"
        del big, halves


def test_gold_cc_mult_batch_of_8_equals_loop():
    """8 gold multiplications under one key in one cc_mult_batch call (d0 / d1 folded into the key-switch sums, the
    (digit, own limb) pairs skipped): every result equals cc_mult's, the fixture's pair the reference engine's digest."""
    from liberate_fhe_amd.fhe import ckks_engine
    rec = GOLD["gold"]
    eng = ckks_engine(devices=["cuda:0"], **rec["params"])
    s = rec["seeds"]
    evk = synth.key_switch_key(eng, s["evk"])
    pairs = [(synth.ciphertext(eng, s["ct_a"], 0), synth.ciphertext(eng, s["ct_b"], 0))]
    pairs += [(synth.ciphertext(eng, 200 + i, 0), synth.ciphertext(eng, 300 + i, 0)) for i in range(7)]
    batch = eng.cc_mult_batch(pairs, evk)
    assert _digest(batch[0]) == rec["ops"]["cc_mult(a,b,evk)"]               # the reference engine's output
    for i, (a, b) in enumerate(pairs):
        one = eng.cc_mult(a, b, evk)
        for comp in range(2):
            assert torch.equal(one.data[comp][0], batch[i].data[comp][0]), (i, comp)
        del one
