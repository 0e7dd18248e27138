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


def _chain_tables(ctx, ntt, total, L):
    """Device tables of the last L limbs of the chain + the host constants the oracle takes for the same limbs."""
    from liberate_fhe_amd.ntt import twiddles
    sl = lambda t: t[0][total - L:]
    psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
    st = torch.cuda.current_stream().cuda_stream
    psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
    h = lambda v: np.asarray(v, dtype=np.int64)[total - L:]
    cs = (h(ctx.q_lower_bits), h(ctx.q_higher_bits), h(ctx.k_lower_bits), h(ctx.k_higher_bits))
    return dict(psi=psi, q2=q2, ql=ql, qh=qh, kl=kl, kh=kh, psi_dp=psi_dp, st=st, cs=cs, q2_h=h(ctx.q_double), Rs_h=h(ctx.R_square))


@pytest.mark.parametrize("LOGN,L,B,ctx_kw", [(16, 30, 70, dict(num_special_primes=4)),
                                              (13, 28, 600, dict(num_scales=23, num_special_primes=4, is_secured=False))])
def test_line_of_record_kernel_through_the_workspace_vs_oracle_and_in_place(LOGN, L, B, ctx_kw):
    """The kernel bench.py times — ntt_pass16_fwd_seq_ws<SKIP0>, reached by lf_ntt_ws from 8 x 4096 tiles on (K.cu:236-323) —
    at the bench shape (logN 16, 30 limbs, B = 70: 33 600 tiles, tile pairs per class multiples of 8 -> XCD-aware block order, a
    block's 8 tiles are 8 consecutive polynomials of one (limb, tile) pair, B % 8 != 0 so blocks straddle pairs and reload
    their last-stage twiddles mid-loop) and at logN 13 with 23 + 5 limbs (pairs not multiples of 8: plain order, a block's
    tiles are polynomials 8 apart; class padding leaves dead tiles).  Words outside [0, 2q) — signed-lazy, exactly 2q,
    arbitrary below 2^61, and any int64 at all — sit in polynomials that are the FIRST, a MIDDLE and the LAST tile of a
    block's loop and on both sides of the integer / fp64 class boundary, so flagged column waves (third plane) are met inside
    the loop beside unflagged neighbours.  Checked: every limb of 8 polynomials word for word against the C oracle; the whole
    batch against the strictly in-place lf_ntt; lf_ntt_pass_ws(1) + (2) against lf_ntt_ws; all of it with the column pass
    taking the tiles' first stage (LF_TUNE_WS_EXTRA_STAGE = 1: seq_ws<true>) and not (0: seq_ws<false>)."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context
    from oracle import oracle as orc
    ctx = ckks_context(logN=LOGN, **ctx_kw)
    ntt = ntt_context(ctx, devices=["cuda:0"])
    total, N = len(ctx.q), ctx.N
    tiles = B * L * (N >> 12)
    assert total >= L and tiles >= 8 * 4096, tiles
    q = np.array(ctx.q[total - L:], dtype=np.int64)
    n_int = int((q >= (1 << 41)).sum())
    assert 0 < n_int < L                                            # both arithmetic classes in one launch
    T = _chain_tables(ctx, ntt, total, L)

    gen = torch.Generator(device="cuda").manual_seed(1600 + LOGN)
    x0 = torch.empty((B, L, N), dtype=torch.int64, device="cuda")
    for r in range(L):
        x0[:, r] = torch.randint(0, 2 * int(q[r]), (B, N), generator=gen, device="cuda", dtype=torch.int64)

    # polynomials by their place in a block's 8-tile loop: XCD-aware order -> tile i of block k is polynomial (8 k + i) % B;
    # plain order (logN 13 shape) -> polynomial (x + 64 k + 8 i) % B for XCD lane x
    xcd_order = ((n_int << (LOGN - 12)) & 7) == 0 and (((L - n_int) << (LOGN - 12)) & 7) == 0
    assert xcd_order == (LOGN == 16)
    first, mid, last = (0, 3, 7) if xcd_order else (0, 24, 56)
    straddle = B - 1                                                # last polynomial of a pair, followed by polynomial 0 of the next
    dirty = [first, mid, last, straddle, B // 2 + 1]               # oracle-defined out-of-range words
    wild = [first + 8 * 2 + 1, B - 2]                               # any int64: the two forms against each other only
    clean_checked = [1, B // 2]                                     # unflagged polynomials beside flagged ones
    rng = np.random.default_rng(77 + LOGN)
    host = {b: x0[b].cpu().numpy() for b in dirty + wild + clean_checked}
    for b in dirty:
        for r in range(L):
            qq = int(q[r])
            cols = rng.integers(0, N, size=6)
            host[b][r, cols[0]] -= 2 * qq                           # signed-lazy (D.4)
            host[b][r, cols[1]] = 2 * qq                            # boundary word
            host[b][r, cols[2:4]] = rng.integers(-(2 ** 61), 2 ** 61, size=2, dtype=np.int64)
            host[b][r, (r * 37) % N] = -1                           # low columns: another wave of the column pass
            host[b][r, N - 1 - r] = 3 * qq
        host[b][L - 1] = x0[b, L - 1].cpu().numpy()                 # one limb of a dirty polynomial stays clean
    for b in wild:
        for r in range(L):
            host[b][r, rng.integers(0, N, size=5)] = rng.integers(-(2 ** 63), 2 ** 63 - 1, size=5, dtype=np.int64)
    for b in dirty + wild:
        x0[b] = torch.from_numpy(host[b]).cuda()

    words = int(lib.lf_ntt_ws_words(B, L, LOGN))
    ws = torch.empty((words,), dtype=torch.int64, device="cuda")
    qh_ = q.ctypes.data

    def in_place(x):
        check(lib.lf_ntt(x.data_ptr(), B, L, LOGN, T["psi"].data_ptr(), T["psi_dp"], qh_, 0, 0, T["q2"].data_ptr(), T["ql"].data_ptr(),
                         T["qh"].data_ptr(), T["kl"].data_ptr(), T["kh"].data_ptr(), 0, T["st"]), "lf_ntt")

    def through_ws(x):
        check(lib.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), B, L, LOGN, T["psi"].data_ptr(), T["psi_dp"], qh_, 0, 0, T["ql"].data_ptr(),
                            T["qh"].data_ptr(), T["kl"].data_ptr(), T["kh"].data_ptr(), 0, T["st"]), "lf_ntt_ws")

    def two_passes(x):
        for which in (1, 2):
            check(lib.lf_ntt_pass_ws(x.data_ptr(), ws.data_ptr(), B, L, LOGN, T["psi"].data_ptr(), T["psi_dp"], qh_, 0, 0, which,
                                     T["ql"].data_ptr(), T["qh"].data_ptr(), T["kl"].data_ptr(), T["kh"].data_ptr(), 0, T["st"]), "lf_ntt_pass_ws")

    ref = x0.clone()
    in_place(ref)
    torch.cuda.synchronize()

    psi_h = np.ascontiguousarray(ctx.psi_br[total - L:].copy())
    orc.mont_enter(psi_h, T["Rs_h"], L, *T["cs"])
    want = {}
    for b in dirty + clean_checked:
        w = host[b].copy()
        orc.ntt(w, psi_h, L, LOGN, T["q2_h"], *T["cs"])
        want[b] = w
        assert (ref[b].cpu().numpy() == w).all(), f"lf_ntt, polynomial {b}: differs from the oracle"

    assert lib.lf_tune(4, -1) == 1
    try:
        for extra in (1, 0):
            assert lib.lf_tune(4, extra) in (0, 1)
            for name, run in (("lf_ntt_ws", through_ws), ("lf_ntt_pass_ws 1 + 2", two_passes)):
                ws.fill_(-1)                                        # a hostile workspace: stale flags raised, stale planes
                x = x0.clone()
                run(x)
                torch.cuda.synchronize()
                for b, w in want.items():
                    got = x[b].cpu().numpy()
                    bad = np.argwhere(got != w)
                    assert bad.size == 0, f"{name}, extra stage {extra}, polynomial {b}: {len(bad)} words differ from the oracle, first at limb/word {bad[0].tolist()}"
                assert torch.equal(x, ref), f"{name}, extra stage {extra}: whole batch differs from lf_ntt in place"
                del x
    finally:
        lib.lf_tune(4, 1)
