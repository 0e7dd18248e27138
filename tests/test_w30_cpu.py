"""The reference's 30-bit / int32 word mode (ckks_context.py:213-216; K.cu templates over scalar_t) on the CPU side: the
oracle's int32 instantiation against big-integer definitions, this package's 30-bit context against the reference's
(when /root/reference is present), and the engine's refusal of a mode the reference's own engine cannot run."""
import random

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import Limbs, pick_primes30
from liberate_fhe_amd.fhe.context.ckks_context import bit_reverse_indices, stage_butterfly_indices

R30 = 1 << 30


def consts(q):
    k = (R30 * pow(R30, -1, q) - 1) // q
    lb = (1 << 15) - 1
    return q & lb, q >> 15, k & lb, k >> 15, k


@pytest.mark.parametrize("q", [16801793, 268369921, 268271617, 33710081])
def test_mm30_is_exact_redc_for_signed_inputs(q):
    """lfo30 mm(a, b) == (a b + ((a b k) mod 2^30) q) / 2^30 exactly for signed lazy operands, redc likewise: the int32
    instantiation of the reference's template (15-bit halves, wrapping int32 arithmetic) computes the closed form."""
    ql, qh, kl, kh, k = consts(q)
    assert (k * q + 1) % R30 == 0
    rnd = random.Random(q)
    for _ in range(20000):
        a, b = rnd.randrange(-2 * q + 1, 2 * q), rnd.randrange(-2 * q + 1, 2 * q)
        x = a * b
        want = (x + ((x * k) % R30) * q) // R30
        assert orc.mm_scalar(a, b, ql, qh, kl, kh, bits=30) == want
        if 0 <= a < 2 * q and 0 <= b < 2 * q:
            assert 0 <= want < 2 * q and (want * R30 - a * b) % q == 0
    for x in [0, 1, q - 1, q, 2 * q - 1] + [rnd.randrange(0, 2 * q) for _ in range(2000)]:
        want = (x + ((x * k) % R30) * q) // R30
        assert orc.redc_scalar(x, ql, qh, kl, kh, bits=30) == want and 0 <= want <= q


@pytest.mark.parametrize("logN", [3, 6, 10])
def test_ntt30_is_evaluation_at_odd_powers_and_inverts(logN):
    lim = Limbs(logN, pick_primes30(logN, 2, 1), bits=30)
    psi, ipsi = lim.mont_tables()
    assert psi.dtype == np.int32
    N = lim.N
    x = lim.uniform(1)
    y = x.copy()
    orc.mont_enter(y, lim.Rs, lim.rows, *lim.mont_args())
    orc.ntt(y, psi, lim.rows, logN, lim._2q, *lim.mont_args())
    brev = bit_reverse_indices(logN)
    for r, q in enumerate(lim.q):
        g = lim.root[r]
        coeffs = [int(v) for v in x[r]]
        for kk in range(0, N, max(1, N // 16)):
            w = pow(g, 2 * int(brev[kk]) + 1, q)
            val = 0
            for c in reversed(coeffs):
                val = (val * w + c) % q
            assert 0 <= y[r, kk] < 2 * q and (int(y[r, kk]) - val * R30) % q == 0
    orc.intt(y, ipsi, lim.Ninv, lim.rows, logN, lim._2q, *lim.mont_args())
    orc.mont_redc(y, lim.rows, *lim.mont_args())
    orc.reduce_2q(y, lim.rows, lim._2q)
    assert (y == x).all()


@pytest.mark.parametrize("logN", [1, 5, 9])
def test_compact_and_table_driven_ntt30_agree(logN):
    lim = Limbs(logN, pick_primes30(logN, 2, 1), bits=30)
    psi, ipsi = lim.mont_tables()
    ev, od, tw = stage_butterfly_indices(logN, False)
    iev, iod, itw = stage_butterfly_indices(logN, True)
    x = lim.uniform(3, lazy=True)
    a, b = x.copy(), x.copy()
    orc.ntt(a, psi, lim.rows, logN, lim._2q, *lim.mont_args())
    orc.ntt_tab(b, ev, od, np.ascontiguousarray(psi[:, tw]), lim.rows, lim._2q, *lim.mont_args())
    assert (a == b).all()
    orc.intt(a, ipsi, lim.Ninv, lim.rows, logN, lim._2q, *lim.mont_args())
    orc.intt_tab(b, iev, iod, np.ascontiguousarray(ipsi[:, itw]), lim.Ninv, lim.rows, lim._2q, *lim.mont_args())
    assert (a == b).all()


def test_context30_equals_the_reference_context():
    """ckks_context(buffer_bit_length = 30): primes, Montgomery constants and twiddles equal the imported reference's."""
    from tests.golden import refdriver
    if not refdriver.reference_available():
        pytest.skip("reference not present (GPU box): the comparison runs in the build container")
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    import torch
    kw = dict(buffer_bit_length=30, scale_bits=24, logN=12, num_scales=4, num_special_primes=2, is_secured=False)
    ref = refdriver.reference_context(**kw)
    mine = ckks_context(**kw)
    assert mine.torch_dtype == ref.torch_dtype == torch.int32 and mine.numpy_dtype == ref.numpy_dtype == np.int32
    for name in ("q", "R", "R_square", "q_lower_bits", "q_higher_bits", "q_double", "R_inv", "k", "k_lower_bits", "k_higher_bits", "N_inv",
                 "half_buffer_bit_length", "lower_bits_mask", "full_bits_mask", "message_bits"):
        assert getattr(mine, name) == getattr(ref, name), name
    assert (np.asarray(mine.forward_psi) == np.asarray(ref.forward_psi)).all()
    assert (np.asarray(mine.backward_psi_inv) == np.asarray(ref.backward_psi_inv)).all()


def test_engine_refuses_the_word_mode_the_reference_engine_cannot_run():
    """The reference's OWN engine breaks in 30-bit mode at key generation (its samplers return int64 words, the constants
    are int32: ckks_engine.py:355 hands both to one kernel template); this engine says so at construction."""
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    with pytest.raises(ValueError, match="62"):
        ckks_engine(devices=["cpu"], backend=OracleBackend(), buffer_bit_length=30, scale_bits=24, logN=12, num_scales=4,
                    num_special_primes=2, is_secured=False)


def _w30_ops_digests(ntt, rec, dev):
    """Every one of the 15 functions once through ntt_context's methods on the fixture's seeded words (tests/golden/
    make_golden.py run_w30 — the same sequence the REFERENCE's ntt_context ran); {op: sha256}."""
    import hashlib
    import torch
    q, N = rec["q"], ntt.ctx.N
    rng = np.random.default_rng(rec["seed"])
    lazy = np.stack([rng.integers(0, 2 * qi, size=N) for qi in q]).astype(np.int32)
    other = np.stack([rng.integers(0, 2 * qi, size=N) for qi in q]).astype(np.int32)
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
    fresh = lambda: torch.from_numpy(lazy.copy()).to(dev)
    got = {}
    for name in ("ntt", "enter_ntt", "intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed", "mont_redc", "reduce_2q",
                 "make_signed", "make_unsigned", "mont_enter"):
        t = fresh()
        getattr(ntt, name)([t], 0, -2)
        got[name] = sha(t)
    a, b = fresh(), torch.from_numpy(other.copy()).to(dev)
    got["mont_mult"] = sha(ntt.mont_mult([a], [b], 0, -2)[0])
    got["mont_add"] = sha(ntt.mont_add([a], [b], 0, -2)[0])
    got["mont_sub"] = sha(ntt.mont_sub([a], [b], 0, -2)[0])
    one = torch.from_numpy(rng.integers(-1, 2, size=N).astype(np.int32)).to(dev)
    assert hashlib.sha256(np.ascontiguousarray(one.cpu().numpy()).tobytes()).hexdigest() == rec["tile_input_sha256"]
    got["tile_unsigned"] = sha(ntt.tile_unsigned([one], 0, -2)[0])
    return got


def test_ntt_context30_over_the_oracle_reproduces_the_reference_digests():
    """This package's ckks_context + ntt_context in 30-bit mode (tables, partition, packs) driving the oracle's int32
    instantiation == the digests the REFERENCE's ntt_context produced over the same oracle (tests/golden/w30_ntt.json)."""
    import json
    import os
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context
    from tests.oracle_backend import make_ops
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "w30_ntt.json")))
    ctx = ckks_context(**rec["params"])
    ntt = ntt_context(ctx, devices=["cpu"], ops=make_ops("w30_checker"))
    assert [int(ctx.q[i]) for i in ntt.p.d_special[0]] == rec["q"]
    assert _w30_ops_digests(ntt, rec, "cpu") == rec["ops"]
