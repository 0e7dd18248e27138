"""SURVEY.md §8(f) rows 2-3 on the product path: key generation, encrypt / decrypt, encode / decode against
fixtures recorded from the REFERENCE engine (tests/golden/make_golden.py keygen -> keygen_encdec.json).

Randomness is injected on both sides through tests.helpers.SeededCsprng (the reference's CSPRNG cannot be seeded,
SURVEY.md §8c), re-seeded at the same points; the operation sequence is one function shared by the generator
and these tests, so the draws line up.  Integer results are compared bit for bit (SHA-256); the three steps that
contain an fp64 FFT are compared with the tolerance written at the assertion.
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from tests.golden.make_golden import integer_plaintext, keygen_sequence, message, reseed, KEYGEN_SEED

warnings.filterwarnings("ignore", category=UserWarning)
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keygen_encdec.json")))


def make(device, rec, n_dev=1, backend=None):
    from liberate_fhe_amd.fhe import ckks_engine
    kw = {} if backend is None else {"backend": backend}
    return ckks_engine(devices=[device] * n_dev, **kw, **rec["params"])


def check_keygen(eng, rec):
    got = keygen_sequence(eng)
    assert set(got) == set(rec["ops"])
    for name in rec["ops"]:   # in recording order: the first mismatch names the first wrong step
        assert got[name] == rec["ops"][name], name


@pytest.mark.parametrize("name", ["keygen_small", "keygen_small_x2", "keygen_bronze"])
def test_checker_engine_keygen_equals_reference(name):
    from tests.oracle_backend import OracleBackend
    rec = GOLD[name]
    check_keygen(make("cpu", rec, rec["n_devices"], OracleBackend()), rec)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["keygen_small", "keygen_small_x2", "keygen_bronze"])
def test_hip_engine_keygen_equals_reference(name):
    """create_secret_key / create_public_key / create_evk / create_rotation_key / create_conjugation_key /
    create_key_switching_key(a = crs) / encrypt / decrypt (double and triplet) with injected a, e, v: every
    tensor equal to the reference engine's (eng.py:350-411, 601-652, 1054-1070, 1157-1232, 417-595)."""
    rec = GOLD[name]
    check_keygen(make("cuda:0", rec, rec["n_devices"]), rec)


def check_encdec(eng, rec):
    N, keep = eng.ctx.N, rec["keep"]
    m = message(N, rec["message_seed"])
    scale = float(eng.scale)
    for level in (0, 2):
        reseed(eng, rec["rng_seed"])
        pt = eng.encode(m, level=level)[0]
        want = np.array(rec[f"encode(level={level})"], dtype=np.int64)
        got = pt[:keep].cpu().numpy()
        # encode = fp64 FFT, x scale x deviation, stochastic rounding with the injected uniform draws.  The FFT
        # libraries (pocketfft on the host that recorded the fixture, rocFFT here) differ in the last bits of the
        # coefficient; a coefficient of magnitude ~2^40 carries an absolute error of ~2^-12, so a draw within that
        # distance of the rounding threshold may fall on the other side: integer plaintexts agree to +-1, and
        # all but a fraction ~2^-9 of them exactly.
        diff = np.abs(got - want)
        assert diff.max() <= 1, level
        assert (diff != 0).mean() < 0.01, level
        back = eng.decode([pt], level=level)[:keep]
        ref = np.array([complex(a, b) for a, b in rec[f"decode(encode,level={level})"]])
        # decode(encode(m)): the rounding noise is ~ sqrt(N)/scale, far above fp64 round-off; both engines agree
        # to that precision, and reproduce m up to the level's scale-drift correction (1e-7 .. 2e-6 relative, eng.py:243-263)
        assert np.abs(back - ref).max() < 64 * np.sqrt(N) / scale
        assert np.abs(back - m[:keep]).max() < 1e-5
    # exact integer plaintext in: the FFT is the only inexact step -> 1e-12 relative to the largest output
    dec = eng.decode([torch.from_numpy(integer_plaintext(N)).to(eng.ntt.devices[0])], level=0)[:keep]
    ref = np.array([complex(a, b) for a, b in rec["decode(integer_plaintext)"]])
    assert np.abs(dec - ref).max() <= 1e-12 * np.abs(ref).max() * np.log2(N)
    # encorypt -> decrode through the bias guard (DC coefficient split off, re-added through a 3-prime CRT)
    reseed(eng, KEYGEN_SEED)
    sk = eng.create_secret_key()
    pk = eng.create_public_key(sk)
    reseed(eng, 5150)
    out = eng.decrode(eng.encorypt(m + 3.0, pk), sk)
    ref = np.array([complex(a, b) for a, b in rec["decrode(encorypt(m+3))"]])
    assert np.abs(out[:keep] - ref).max() < 256 * np.sqrt(N) / scale       # same noise level as the reference run
    assert np.abs(out - (m + 3.0)).max() < max(4 * rec["decrode_max_err"], 256 * np.sqrt(N) / scale)


@pytest.mark.parametrize("name", ["encdec_small"])
def test_checker_engine_encode_decode_equals_reference(name):
    from tests.oracle_backend import OracleBackend
    rec = GOLD[name]
    check_encdec(make("cpu", rec, 1, OracleBackend()), rec)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["encdec_small", "encdec_silver"])
def test_hip_engine_encode_decode_equals_reference(name):
    """encode / decode / bias guard / decrypt tail (encdec.py:9-323, eng.py:1472-1681) on cuda:0 vs vectors the
    reference produced; tolerances are stated in check_encdec."""
    rec = GOLD[name]
    check_encdec(make("cuda:0", rec), rec)
