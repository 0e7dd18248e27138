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
