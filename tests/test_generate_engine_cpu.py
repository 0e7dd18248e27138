"""The reference's own engine test (src/liberate/fhe/tests/test_generate_engine.py): an engine is built
for every (logN, scale_bits) in {14,15,16} x {20,25,..,45} with the default security settings.  The
reference's shipped prime tables have too few 20-bit (and, at logN 16, 25-bit) NTT primes for those
rings, where its constructor raises NotEnoughPrimes; the expected chain lengths below were recorded from
the reference's ckks_context (and are re-checked live against it where /root/reference exists)."""
import warnings

import pytest

from tests.golden import refdriver as rd

warnings.filterwarnings("ignore", category=UserWarning)

# (logN, scale_bits) -> number of primes in the chain, or None where the reference raises NotEnoughPrimes
EXPECTED = {
    (14, 20): None, (14, 25): 12, (14, 30): 10, (14, 35): 9, (14, 40): 8, (14, 45): 8,
    (15, 20): None, (15, 25): 28, (15, 30): 24, (15, 35): 21, (15, 40): 19, (15, 45): 17,
    (16, 20): None, (16, 25): None, (16, 30): 52, (16, 35): 45, (16, 40): 40, (16, 45): 35,
}


def ctx_params(logN, scale_bits):
    return dict(logN=logN, scale_bits=scale_bits, security_bits=128, num_scales=None, num_special_primes=2,
                buffer_bit_length=62, sigma=3.2, uniform_ternary_secret=True, quantum="post_quantum",
                distribution="uniform")


@pytest.mark.parametrize("logN,scale_bits", sorted(EXPECTED))
def test_make_engine(logN, scale_bits):
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.presets import errors
    from tests.oracle_backend import OracleBackend
    want = EXPECTED[logN, scale_bits]
    if want is None:
        with pytest.raises(errors.NotEnoughPrimes):
            ckks_engine(devices=["cpu"], backend=OracleBackend(), **ctx_params(logN, scale_bits))
        return
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), **ctx_params(logN, scale_bits))
    assert isinstance(eng, ckks_engine)
    assert len(eng.ctx.q) == want and eng.num_levels == want - 3
    assert all(q % (2 * eng.ctx.N) == 1 for q in eng.ctx.q)


@pytest.mark.reference
@pytest.mark.skipif(not rd.reference_available(), reason="/root/reference not present")
@pytest.mark.parametrize("logN,scale_bits", sorted(EXPECTED))
def test_chain_equals_reference_context(logN, scale_bits):
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    outcome = []
    for make in (lambda: rd.reference_context(**ctx_params(logN, scale_bits)),
                 lambda: ckks_context(**ctx_params(logN, scale_bits))):
        try:
            outcome.append([int(x) for x in make().q])
        except Exception as e:
            outcome.append(type(e).__name__)
    assert outcome[0] == outcome[1]
    assert (None if isinstance(outcome[1], str) else len(outcome[1])) == EXPECTED[logN, scale_bits]


@pytest.mark.reference
@pytest.mark.skipif(not rd.reference_available(), reason="/root/reference not present")
def test_scale_prime_pools_equal_shipped_table():
    """Every entry of the reference's scale_primes.pkl — 240 (scale_bits, N) pairs, including the truncated
    pools and the two "not enough primes" entries — is reproduced by the on-demand generator."""
    import pickle
    from liberate_fhe_amd.fhe.context import primes
    table = pickle.load(open("/root/reference/src/liberate/fhe/cache/resources/scale_primes.pkl", "rb"))
    assert len(table) == 240
    for (sb, N), want in table.items():
        try:
            got = list(primes.scale_prime_pool(sb, N))
        except LookupError:
            got = None
        assert got == (None if isinstance(want, str) else want), (sb, N)
