"""This repo's engine (orchestration + checker backend, on CPU) against the REAL reference engine driven
by the same oracle kernels, on identical tensors.  Runs only where /root/reference exists (the build
container); the GPU box relies on the golden digests these runs produced (tests/golden/*.json)."""
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


def same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)


@pytest.mark.parametrize("n_dev", [1, 2, 3])
def test_precomputed_tables_match(n_dev):
    ref, me = rd.reference_engine(n_dev, **SMALL), mine(n_dev, **SMALL)
    assert ref.hash == me.hash and ref.num_levels == me.num_levels
    for name in ("len_devices", "neighbor_devices", "parts_alloc", "stor_ids", "deviations", "corrections",
                 "final_q_ind", "galois_deltas"):
        assert getattr(ref, name) == getattr(me, name), name
    for lvl in range(ref.num_levels):
        same(ref.rescale_scales[lvl], me.rescale_scales[lvl])
        for P_ind in range(ref.ntt.num_special_primes):
            same(ref.PiRs[lvl][P_ind], me.PiRs[lvl][P_ind])
    same(ref.mont_PR, me.mont_PR)
    same(ref.final_scalar, me.final_scalar)
    # ntt_context packs: same keys, same tensors (twiddles compared through their compact form)
    for dev in range(n_dev):
        assert set(ref.ntt.parts_pack[dev]) == set(me.ntt.parts_pack[dev])
        for key, item in ref.ntt.parts_pack[dev].items():
            other = me.ntt.parts_pack[dev][key]
            for f in ("Rs", "Rs_scale", "_2q"):
                same(item[f], other[f])
            for f in ("Y_scalar",):
                if f in item:
                    assert (item[f] is None) == (other[f] is None)
                    if item[f] is not None:
                        assert torch.equal(item[f], other[f])
            if item.get("L_scalar") is not None:
                same(item["L_scalar"], other["L_scalar"])
            if "L_enter" in item:
                for a, b in zip(item["L_enter"], other["L_enter"]):
                    assert (a is None) == (b is None)
                    if a is not None:
                        same(a, b)


@pytest.mark.parametrize("n_dev", [1, 2])
def test_hot_path_matches_reference_engine(n_dev):
    from liberate_fhe_amd.utils import synth
    ref, me = rd.reference_engine(n_dev, **SMALL), mine(n_dev, **SMALL)
    # real keys and ciphertexts made by the reference engine, handed to both engines
    sk = ref.create_secret_key()
    pk = ref.create_public_key(sk)
    evk = ref.create_evk(sk)
    rotk = ref.create_rotation_key(sk, 5)
    conjk = ref.create_conjugation_key(sk)
    np.random.seed(1)
    m1, m2 = ref.example(-1, 1), ref.example(-1, 1)
    c1, c2 = ref.encorypt(m1, pk), ref.encorypt(m2, pk)

    def both(fn):
        r, m = fn(ref), fn(me)
        assert r.level == m.level and r.origin == m.origin
        for x, y in zip(r.data, m.data):
            same(x, y)
        return r

    both(lambda e: e.rescale(c1))
    prod = both(lambda e: e.cc_mult(c1, c2, evk))
    both(lambda e: e.cc_mult(prod, prod, evk))
    both(lambda e: e.rotate_single(c1, rotk))
    both(lambda e: e.rotate_single(prod, rotk))
    both(lambda e: e.conjugate(c1, conjk))
    both(lambda e: e.cc_add(c1, c2))
    both(lambda e: e.cc_sub(c1, c2))
    both(lambda e: e.level_up(c1, 3))
    err = np.abs(me.decrode(me.cc_mult(c1, c2, evk), sk) - m1 * m2).max()
    assert err < 1e-7
    assert np.abs(ref.decrode(c1, sk) - me.decrode(c1, sk)).max() == 0

    # synthetic inputs (what the GPU-box fixtures use)
    for e in (ref, me):
        e._ct = synth.ciphertext(e, 11, 0), synth.ciphertext(e, 12, 0)
        e._k = synth.key_switch_key(e, 13), synth.key_switch_key(e, 14, origin="rotation key:7")
    both(lambda e: e.cc_mult(e._ct[0], e._ct[1], e._k[0]))
    both(lambda e: e.rotate_single(e._ct[0], e._k[1]))


def test_own_keygen_and_roundtrip_decodes():
    me = mine(1, **SMALL)
    sk = me.create_secret_key()
    pk = me.create_public_key(sk)
    evk = me.create_evk(sk)
    rotk = me.create_rotation_key(sk, 3)
    gk = me.create_galois_key(sk)
    np.random.seed(2)
    m1, m2 = me.example(-1, 1), me.example(-1, 1)
    c1, c2 = me.encorypt(m1, pk), me.encorypt(m2, pk)
    assert np.abs(me.decrode(c1, sk) - m1).max() < 1e-8
    assert np.abs(me.decrode(me.cc_mult(c1, c2, evk), sk) - m1 * m2).max() < 1e-7
    assert np.abs(me.decrode(me.rotate_single(c1, rotk), sk) - np.roll(m1, 3)).max() < 1e-8
    assert np.abs(me.decrode(me.rotate_galois(c1, gk, 37), sk) - np.roll(m1, 37)).max() < 1e-7
    assert np.abs(me.decrode(me.mult(c1, me.cc_mult(c1, c2, evk), evk), sk) - m1 * m1 * m2).max() < 1e-6
    pt = me.encode(m1)
    assert np.abs(me.decode(me.decrypt(me.encrypt(pt, pk), sk)) - m1).max() < 1e-8
