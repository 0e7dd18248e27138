"""The evaluator surface around the hot path (scalar / plaintext operands, negation, statistics, data
movement, save / load, multiparty helpers) against the REAL reference engine on identical tensors and
identical random draws.  Build container only; the GPU box uses the digests recorded by
tests/golden/make_golden.py ("evaluator" section)."""
import os
import warnings

import numpy as np
import pytest
import torch

from tests.golden import refdriver as rd

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not rd.reference_available(), reason="/root/reference not present")]

warnings.filterwarnings("ignore", category=UserWarning)
SMALL = dict(logN=12, num_scales=5, num_special_primes=2, is_secured=False)


def mine(n_dev=1, **params):
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    return ckks_engine(devices=["cpu"] * n_dev, backend=OracleBackend(), **params)


def reseed(*engines, seed=7):
    """Give every engine the same deterministic sampler, positioned at the same point of its stream."""
    for e in engines:
        e.rng = rd.SeededCsprng(e.ctx.N, [len(d) for d in e.ntt.p.d], 2, devices=e.ntt.devices, seed=seed)


def same_struct(r, m):
    assert r.level == m.level and r.origin == m.origin
    assert (r.include_special, r.ntt_state, r.montgomery_state) == (m.include_special, m.ntt_state, m.montgomery_state)
    if hasattr(r.data[0], "origin"):
        assert len(r.data) == len(m.data)
        for x, y in zip(r.data, m.data):
            same_struct(x, y)
        return
    assert len(r.data) == len(m.data)
    for x, y in zip(r.data, m.data):
        assert len(x) == len(y)
        for s, t in zip(x, y):
            assert s.shape == t.shape and torch.equal(s, t)


DEEP = dict(SMALL, num_scales=10)


def make_world(n_dev, params):
    ref, me = rd.reference_engine(n_dev, **params), mine(n_dev, **params)
    sk = ref.create_secret_key()
    pk = ref.create_public_key(sk)
    evk = ref.create_evk(sk)
    gk = ref.create_galois_key(sk)
    np.random.seed(1)
    m1, m2 = ref.example(-1, 1), ref.example(-1, 1)
    c1, c2 = ref.encorypt(m1, pk), ref.encorypt(m2, pk)
    return dict(ref=ref, me=me, sk=sk, pk=pk, evk=evk, gk=gk, m1=m1, m2=m2, c1=c1, c2=c2)


@pytest.fixture(scope="module", params=[1, 2])
def world(request):
    return make_world(request.param, SMALL)


@pytest.fixture(scope="module")
def deep_world():
    return make_world(1, DEEP)


def both(w, fn, seed=7):
    reseed(w["ref"], w["me"], seed=seed)
    r, m = fn(w["ref"]), fn(w["me"])
    same_struct(r, m)
    return r


def test_scalar_operands(world):
    w, c1 = world, world["c1"]
    both(w, lambda e: e.negate(c1))
    both(w, lambda e: e.mult_int_scalar(c1, -7))
    both(w, lambda e: e.mult(3, c1))
    both(w, lambda e: e.mult_scalar(c1, 0.37))
    both(w, lambda e: e.mult(c1, -1.5))
    both(w, lambda e: e.add_scalar(c1, 1.25))
    both(w, lambda e: e.sub_scalar(c1, 0.5))
    both(w, lambda e: e.add(2, c1))
    both(w, lambda e: e.sub(2.5, c1))
    both(w, lambda e: e.reduce_error(c1))
    both(w, lambda e: e.rescale(c1, exact_rounding=False))
    with pytest.raises(Exception, match="Unsupported data types"):
        w["me"].mult("x", c1)


def test_plaintext_operands(world):
    w, c1, m2 = world, world["c1"], world["m2"]
    prod = both(w, lambda e: e.mc_mult(m2, c1))
    both(w, lambda e: e.mult(c1, m2))
    both(w, lambda e: e.mult(list(m2), c1))
    tot = both(w, lambda e: e.mc_add(m2, c1))
    both(w, lambda e: e.mc_sub(m2, c1))
    both(w, lambda e: e.cm_sub(c1, m2))
    both(w, lambda e: e.add(c1, m2))
    me, sk = w["me"], w["sk"]
    assert np.abs(me.decrode(prod, sk) - w["m1"] * m2).max() < 1e-6
    assert np.abs(me.decrode(tot, sk) - (w["m1"] + m2)).max() < 1e-6


def test_statistics_and_powers(deep_world):
    world = deep_world
    w, c1, c2, evk, gk = world, world["c1"], world["c2"], world["evk"], world["gk"]
    both(w, lambda e: e.sum(c1, gk))
    mean = both(w, lambda e: e.mean(c1, gk))
    both(w, lambda e: e.square(c1, evk))
    both(w, lambda e: e.square(c1, evk, relin=False))
    both(w, lambda e: e.pow(c1, 3, evk))
    both(w, lambda e: e.pow(c1, 4, evk))
    both(w, lambda e: e.var(c1, evk, gk))
    both(w, lambda e: e.cov(c1, c2, evk, gk))
    both(w, lambda e: e.sqrt(e.mult_scalar(c1, 0.5), evk, e=0.7))
    got = w["me"].decrode(mean, w["sk"])
    assert np.abs(got - w["m1"].mean()).max() < 1e-6


def test_move_save_load(world, tmp_path):
    w, ref, me, c1, evk = world, world["ref"], world["me"], world["c1"], world["evk"]
    # host form: one natural-order tensor per component
    for text in (c1, evk, w["gk"]):
        same_host(_cpu_ref(ref, text), me.cpu(text), len(ref.ctx.q))
        back = me.cuda(me.cpu(text))
        same_struct(text, back)
    # files written by either side are read by the other
    f1, f2 = str(tmp_path / "ref.pkl"), str(tmp_path / "mine.pkl")
    _save_ref(ref, c1, f1)
    same_struct(c1, me.load(f1))
    me.save(evk, f2)
    same_struct(evk, _load_ref(ref, f2))
    assert me.device(me.load(f2, move_to_gpu=False)) == "cpu"


def same_host(r, m, n_primes):
    """Host forms agree on the meaningful rows (with several GPUs the reference leaves uninitialised rows
    behind the replicated special rows, eng.py:1803-1808)."""
    if hasattr(r.data[0], "origin"):
        for x, y in zip(r.data, m.data):
            same_host(x, y, n_primes)
        return
    for x, y in zip(r.data, m.data):
        assert len(x) == len(y) == 1 and x[0].shape == y[0].shape
        rows = min(x[0].size(0), n_primes - r.level)
        assert torch.equal(x[0][:rows], y[0][:rows])


def _cpu_ref(ref, text):
    """The reference insists on CUDA tensors before downloading; on the CPU stand-in relax that one check."""
    import liberate.fhe.ckks_engine  # noqa: F401
    orig = ref.download_to_cpu

    def download(gpu_data, level, include_special):
        class _T:
            pass
        fake = [t.clone() for t in gpu_data]
        for t in fake:
            t.__class__ = _CudaLike
        return orig(fake, level, include_special)
    ref.download_to_cpu = download
    try:
        return ref.move_to(text, "gpu2cpu")
    finally:
        ref.download_to_cpu = orig


class _Dev:
    type = "cuda"


class _CudaLike(torch.Tensor):
    @property
    def device(self):
        return _Dev()


def _save_ref(ref, text, filename):
    import pickle
    host = _cpu_ref(ref, text)
    with open(filename, "wb") as f:
        pickle.dump(host, f)


def _load_ref(ref, filename):
    return ref.load(filename, move_to_gpu=True)


def test_multiparty_protocol(world):
    """Two parties on engines with identical random draws: every protocol message and the final keys agree
    word for word, and the collectively keyed ciphertext decrypts."""
    w = world
    ref, me = w["ref"], w["me"]
    out = []
    for e in (ref, me):
        reseed(e, seed=21)
        sks = [e.create_secret_key(), e.create_secret_key()]
        pk0 = e.multiparty_create_public_key(sks[0])
        crs = e.multiparty_public_crs(pk0)
        pk1 = e.multiparty_create_public_key(sks[1], a=crs)
        cpk = e.multiparty_create_collective_public_key([pk0, pk1])
        np.random.seed(5)
        m = e.example(-1, 1)
        ct = e.encorypt(m, cpk)
        head = e.multiparty_decrypt_head(ct, sks[0])
        part = e.multiparty_decrypt_partial(ct, sks[1])
        dec = e.multiparty_decrypt_fusion([head, part], level=ct.level)
        assert np.abs(dec - m).max() < 1e-6
        # collective rotation key
        r0 = e.multiparty_create_rotation_key(sks[0], 3)
        r1 = e.multiparty_create_rotation_key(sks[1], 3, a=e.generate_rotation_crs(r0))
        crot = e.multiparty_generate_rotation_key([r0, r1])
        rot = e.rotate_single(ct, crot)
        head = e.multiparty_decrypt_head(rot, sks[0])
        part = e.multiparty_decrypt_partial(rot, sks[1])
        assert np.abs(e.multiparty_decrypt_fusion([head, part], level=rot.level) - np.roll(m, 3)).max() < 1e-5
        # collective evaluation key
        shares = [e.create_key_switching_key(sks[0], sks[0])]
        shares.append(e.create_key_switching_key(sks[1], sks[1], a=e.generate_rotation_crs(shares[0])))
        evk_sum = e.multiparty_sum_evk_share(shares)
        mults = [e.multiparty_mult_evk_share_sum(evk_sum, s) for s in sks]
        cevk = e.multiparty_sum_evk_share_mult(mults)
        sq = e.cc_mult(ct, ct, cevk)
        head = e.multiparty_decrypt_head(sq, sks[0])
        part = e.multiparty_decrypt_partial(sq, sks[1])
        assert np.abs(e.multiparty_decrypt_fusion([head, part], level=sq.level) - m * m).max() < 1e-4
        out.append((cpk, ct, crot, rot, cevk, sq))
    names = ("cpk", "ct", "crot", "rot", "cevk", "sq")
    for name, r, m_ in zip(names, *out):
        if name in ("crot", "rot") and len(ref.ntt.devices) > 1:
            # The reference sums the rotation-key shares on GPU 0 only (eng.py:2589-2595), so with several GPUs
            # its collective key (and what is rotated with it) is wrong on the other GPUs' rows; this build
            # sums on every GPU.  GPU 0's rows agree.
            same_gpu0(r, m_)
        else:
            same_struct(r, m_)


def same_gpu0(r, m):
    if hasattr(r.data[0], "origin"):
        for x, y in zip(r.data, m.data):
            same_gpu0(x, y)
        return
    for x, y in zip(r.data, m.data):
        assert torch.equal(x[0], y[0])
