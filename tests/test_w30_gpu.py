"""The 30-bit / int32 word mode on the GPU: the lf30_* entries of the C ABI (csrc/ckks_w30.hip) through the ntt_cuda shim —
bit-exact against the oracle's int32 instantiation and against the digests recorded from the REFERENCE's ntt_context."""
import json
import os

import numpy as np
import pytest

from tests.helpers import Limbs, pick_primes30

pytestmark = pytest.mark.gpu


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")


@pytest.mark.parametrize("logN", [1, 4, 9, 12, 14])
def test_w30_transforms_and_chains_equal_the_oracle(logN):
    from oracle import oracle as orc
    from liberate_fhe_amd.ntt import ntt_cuda
    lim = Limbs(logN, pick_primes30(logN, 2, 1), bits=30)
    psi, ipsi = lim.mont_tables()
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    q2, Rs, Ninv = dev(lim._2q), dev(lim.Rs), dev(lim.Ninv)
    dpsi, dipsi = dev(psi), dev(ipsi)
    x = lim.uniform(7, lazy=True)
    # ntt / enter_ntt
    for enter in (False, True):
        want = x.copy()
        if enter:
            orc.mont_enter(want, lim.Rs, lim.rows, *lim.mont_args())
        orc.ntt(want, psi, lim.rows, logN, lim._2q, *lim.mont_args())
        got = dev(x)
        if enter:
            ntt_cuda.enter_ntt([got], [Rs], None, None, [dpsi], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
        else:
            ntt_cuda.ntt([got], None, None, [dpsi], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
        assert (got.cpu().numpy() == want).all(), ("enter_ntt" if enter else "ntt")
    # the four inverse chains
    for tail, name in enumerate(("intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed")):
        want = x.copy()
        orc.intt(want, ipsi, lim.Ninv, lim.rows, logN, lim._2q, *lim.mont_args())
        if tail >= 1:
            orc.mont_redc(want, lim.rows, *lim.mont_args())
        if tail >= 2:
            orc.reduce_2q(want, lim.rows, lim._2q)
        if tail >= 3:
            orc.make_signed(want, lim.rows, lim._2q)
        got = dev(x)
        getattr(ntt_cuda, name)([got], None, None, [dipsi], [Ninv], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
        assert (got.cpu().numpy() == want).all(), name
    # round trip
    got = dev(lim.uniform(8))
    keep = got.clone()
    ntt_cuda.enter_ntt([got], [Rs], None, None, [dpsi], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
    ntt_cuda.intt_exit_reduce([got], None, None, [dipsi], [Ninv], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
    assert (got == keep).all()


def test_w30_elementwise_ops_equal_the_oracle_on_signed_and_boundary_words():
    import torch
    from oracle import oracle as orc
    from liberate_fhe_amd.ntt import ntt_cuda
    logN = 10
    lim = Limbs(logN, pick_primes30(logN, 2, 2), bits=30)
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    q2, Rs = dev(lim._2q), dev(lim.Rs)
    rng = np.random.default_rng(11)
    a = np.stack([rng.integers(-2 * q + 1, 2 * q, size=lim.N) for q in lim.q]).astype(np.int32)    # signed-lazy operands
    b = np.stack([rng.integers(0, 2 * q, size=lim.N) for q in lim.q]).astype(np.int32)
    for r, q in enumerate(lim.q):
        a[r, :6] = [0, 1, q - 1, q, 2 * q - 1, -(2 * q - 1)]
        b[r, :6] = [2 * q - 1, q, q - 1, 1, 0, 2 * q - 1]
    want = np.empty_like(a)
    orc.mont_mult(a, b, want, lim.rows, *lim.mont_args())
    got = ntt_cuda.mont_mult([dev(a)], [dev(b)], [c[0]], [c[1]], [c[2]], [c[3]])[0]
    assert got.dtype == torch.int32 and (got.cpu().numpy() == want).all()
    want = a.copy()
    orc.mont_enter(want, lim.Rs, lim.rows, *lim.mont_args())
    got = dev(a)
    ntt_cuda.mont_enter([got], [Rs], [c[0]], [c[1]], [c[2]], [c[3]])
    assert (got.cpu().numpy() == want).all()
    want = b.copy()
    orc.mont_redc(want, lim.rows, *lim.mont_args())
    got = dev(b)
    ntt_cuda.mont_redc([got], [c[0]], [c[1]], [c[2]], [c[3]])
    assert (got.cpu().numpy() == want).all()
    for name in ("reduce_2q", "make_signed", "make_unsigned"):
        want = b.copy()
        getattr(orc, name)(want, lim.rows, lim._2q)
        got = dev(b)
        getattr(ntt_cuda, name)([got], [q2])
        assert (got.cpu().numpy() == want).all(), name
    for name in ("mont_add", "mont_sub"):
        want = np.empty_like(b)
        getattr(orc, name)(b, np.ascontiguousarray(b[:, ::-1].copy()), want, lim.rows, lim._2q)
        got = getattr(ntt_cuda, name)([dev(b)], [dev(np.ascontiguousarray(b[:, ::-1].copy()))], [q2])[0]
        assert (got.cpu().numpy() == want).all(), name
    one = rng.integers(-1, 2, size=lim.N).astype(np.int32)
    want = np.empty((lim.rows, lim.N), dtype=np.int32)
    orc.tile_unsigned(one, want, lim.rows, lim._2q)
    got = ntt_cuda.tile_unsigned([dev(one)], [q2])[0]
    assert (got.cpu().numpy() == want).all()
    # mixed word modes are refused, not reinterpreted (the reference would read int32 constants as int64 words)
    with pytest.raises(TypeError):
        ntt_cuda.reduce_2q([dev(b.astype(np.int64))], [q2])
    # .. by the transforms too: int64 data with this mode's int32 constants (they would be read as int64 words)
    wide = dev(b.astype(np.int64))
    psi32 = dev(lim.mont_tables()[0])
    c32 = [[dev(v)] for v in (lim._2q, lim.ql, lim.qh, lim.kl, lim.kh)]
    for call in (lambda: ntt_cuda.ntt([wide], [None], [None], [psi32], *c32),
                 lambda: ntt_cuda.enter_ntt([wide], [dev(lim.Rs)], [None], [None], [psi32], *c32),
                 lambda: ntt_cuda.intt_exit_reduce([wide], [None], [None], [psi32], [dev(lim.Ninv)], *c32)):
        with pytest.raises(TypeError):
            call()


def test_ntt_context30_on_the_gpu_reproduces_the_reference_digests():
    """ckks_context + ntt_context in 30-bit mode on the HIP shim == the digests recorded from the REFERENCE's ntt_context
    (tests/golden/w30_ntt.json): all 15 functions, twiddles entered into Montgomery form on the device."""
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context
    from tests.test_w30_cpu import _w30_ops_digests
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "w30_ntt.json")))
    ctx = ckks_context(**rec["params"])
    ntt = ntt_context(ctx, devices=["cuda:0"])
    import torch
    assert ntt.psi[0].dtype == torch.int32
    assert _w30_ops_digests(ntt, rec, "cuda:0") == rec["ops"]
