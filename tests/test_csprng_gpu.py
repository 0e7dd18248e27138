"""GPU parity of the sampler kernels (through the C ABI / the reference-named shims) against the numpy
oracle, bit-exact, and of the product Csprng class against streams recorded from the reference's class."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import csprng_oracle as co
from tests import csprng_streams as cs

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "csprng_golden.json")))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def random_states(n, seed):
    g = np.random.default_rng(seed)
    s = g.integers(0, 1 << 32, size=(n, 16), dtype=np.int64)
    if n > 3:
        s[0, 12:14] = (0xFFFFFFFF, 0xFFFFFFFF)
        s[1, 12:14] = (0xFFFFFFF0, 7)
        s[2] = 0
        s[3] = 0xFFFFFFFF
    return s


def tree():
    bt = np.array([[int(lo), int(hi)] for lo, hi in GOLD["cdt"]["btree_low_high"]], dtype=np.uint64)
    return np.ascontiguousarray(bt.T.ravel()), GOLD["cdt"]["size"], GOLD["cdt"]["depth"]


@pytest.mark.parametrize("n", [0, 1, 63, 256, 1000, 65536 + 17])
def test_chacha20(n):
    from liberate_fhe_amd.csprng import chacha20_cuda
    s = random_states(n, n)
    step = 3 * (1 << 31) + 12345
    d = dev(s)
    out = chacha20_cuda.chacha20([d], step)[0]
    assert out.shape == d.shape
    assert np.array_equal(out.cpu().numpy(), co.chacha20_block(s).reshape(n, 16))
    co.step_counter(s, step)
    assert np.array_equal(d.cpu().numpy(), s)


def test_chacha20_golden_blocks():
    from liberate_fhe_amd.csprng import chacha20_cuda
    g = GOLD["chacha20"]
    d = dev(np.array(g["states"], dtype=np.int64))
    out = chacha20_cuda.chacha20([d], g["step"])[0]
    assert out.cpu().tolist() == g["blocks"] and d.cpu().tolist() == g["stepped"]


@pytest.mark.parametrize("channels,L", [(1, 1), (3, 250), (5, 4096), (130, 64)])
def test_randint_fast_and_plain(channels, L):
    from liberate_fhe_amd.csprng import randint_cuda, chacha20_cuda
    s = random_states(channels * L, 11).reshape(channels, L, 16)
    q = np.array([cs.STREAM_Q[i % len(cs.STREAM_Q)] for i in range(channels)], dtype=np.uint64)
    q[0] = (1 << 64) - 1
    step = 1 << 33
    for shift in (0, -1):
        d, ref_states = dev(s), s.copy()
        got = randint_cuda.randint_fast([d], [q], shift, step)[0]
        want = co.randint_fast(ref_states, q, shift, step)
        assert np.array_equal(got.cpu().numpy(), want)
        assert np.array_equal(d.cpu().numpy(), ref_states)
    # unfused: random words first, then the in-place map; the host-pointer form of q as the reference passes it
    d = dev(s)
    rb = chacha20_cuda.chacha20([d.view(-1, 16)], step)[0].view(channels, L, 16)
    want = co.randint(rb.cpu().numpy().copy(), q)
    randint_cuda.randint([rb], [q.__array_interface__["data"][0]])
    assert np.array_equal(rb.cpu().numpy(), want)


@pytest.mark.parametrize("n", [1, 100, 4096, 100003])
def test_discrete_gaussian_fast_and_plain(n):
    from liberate_fhe_amd.csprng import discrete_gaussian_cuda, chacha20_cuda
    flat, size, depth = tree()
    ptr = flat.__array_interface__["data"][0]
    s = random_states(n, 21)
    d, ref_states = dev(s), s.copy()
    got = discrete_gaussian_cuda.discrete_gaussian_fast([d], ptr, size, depth, 12345)[0]
    want = co.discrete_gaussian_fast(ref_states, flat, size, depth, 12345)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(d.cpu().numpy(), ref_states)
    rb = chacha20_cuda.chacha20([dev(s)], 1)[0]
    want = co.discrete_gaussian(rb.cpu().numpy().copy(), flat, size, depth)
    discrete_gaussian_cuda.discrete_gaussian([rb], ptr, size, depth)
    assert np.array_equal(rb.cpu().numpy(), want)


def test_gaussian_tree_boundaries():
    """Random words placed exactly on / next to table entries take the same branch as the oracle."""
    from liberate_fhe_amd.csprng import discrete_gaussian_cuda
    from liberate_fhe_amd.csprng.discrete_gaussian_sampler import cumulative_table
    flat, size, depth = tree()
    table, _ = cumulative_table()
    rb = np.zeros((32, 16), dtype=np.int64)
    for i in range(1, 32):
        for j, delta in enumerate((-1, 0, 1, 0)):
            v = table[i] + delta
            hi, lo = ((v >> 64) << 1) | (j & 1), v & ((1 << 64) - 1)
            rb[i, 4 * j:4 * j + 4] = [lo >> 32, lo & 0xFFFFFFFF, hi >> 32, hi & 0xFFFFFFFF]
    d = dev(rb)
    discrete_gaussian_cuda.discrete_gaussian([d], flat.__array_interface__["data"][0], size, depth)
    got = d.cpu().numpy()
    want = co.discrete_gaussian(rb.copy(), flat, size, depth)
    assert np.array_equal(got, want)
    assert [abs(int(got[i, 4])) for i in range(1, 32)] == list(range(1, 32))      # on the entry -> that index
    assert [abs(int(got[i, 0])) for i in range(1, 32)] == list(range(0, 31))      # one below -> previous index


def test_randround():
    from liberate_fhe_amd.csprng import randround_cuda
    n = 100000
    c = cs.randround_input(n)
    g = np.random.default_rng(3)
    r = g.integers(0, 1 << 32, size=n, dtype=np.int64)
    r[:4] = [0, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000]
    d = dev(r)
    randround_cuda.randround([dev(c)], [d])
    assert np.array_equal(d.cpu().numpy(), co.randround(c, r.copy()))
    with pytest.raises(ValueError):
        randround_cuda.randround([dev(c[:10])], [d])


def test_bad_arguments():
    from liberate_fhe_amd.csprng import chacha20_cuda, discrete_gaussian_cuda
    from liberate_fhe_amd._native import HipError
    with pytest.raises(RuntimeError):
        chacha20_cuda.chacha20([torch.zeros((4, 16), dtype=torch.int64)], 1)
    with pytest.raises(TypeError):
        chacha20_cuda.chacha20([torch.zeros((4, 16), dtype=torch.int32, device="cuda")], 1)
    flat, size, depth = tree()
    with pytest.raises(HipError):
        discrete_gaussian_cuda.discrete_gaussian_fast([torch.zeros((4, 16), dtype=torch.int64, device="cuda")],
                                                     flat.__array_interface__["data"][0], size, depth + 1, 1)


def replay(r, cfg):
    out = []
    for label, method, kw in cs.stream_calls(cfg):
        out.append((label, [t for t in getattr(r, method)(**kw) if t is not None]))
    rr = r.randround(torch.from_numpy(cs.randround_input(cfg["num_coefs"])).to(r.devices[r.local_ids[0]]))
    out.append(("randround", [rr]))
    out.append(("final_states", [s for s in r.states if s is not None]))
    return out


@pytest.mark.parametrize("name", ["small", "gold_like"])
def test_class_streams_match_reference_class(name):
    """Product Csprng on the HIP kernels (logical GPUs all on cuda:0) against the digests recorded from the
    reference's Csprng class."""
    from liberate_fhe_amd.csprng import Csprng
    entry = GOLD["streams"]["configs"][name]
    cfg = entry["config"]
    r = Csprng(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"], devices=["cuda:0"] * cfg["n_dev"],
               seed=GOLD["streams"]["key"], nonce=GOLD["streams"]["nonce"], reference_counter_layout=True)
    for (label, res), want in zip(replay(r, cfg), entry["calls"]):
        assert label == want["label"]
        assert cs.digest([t.cpu() for t in res]) == want["sha256"], label
        assert [t.cpu().ravel()[:len(h)].tolist() for t, h in zip(res, want["head"])] == want["head"] or label == "final_states"


def test_class_default_layout_and_ranks():
    from liberate_fhe_amd.csprng import Csprng
    from tests.oracle_csprng import oracle_csprng_class
    cfg = cs.STREAM_CONFIGS["small"]
    args = (cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"])
    kw = dict(seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE)
    whole = oracle_csprng_class()(*args, devices=["cpu"] * 2, **kw)
    ranks = [Csprng(*args, devices=["cuda:0"] * 2, local_ids=[d], **kw) for d in range(2)]
    for label, method, kw2 in cs.stream_calls(cfg):
        a = getattr(whole, method)(**kw2)
        for d, r in enumerate(ranks):
            b = getattr(r, method)(**kw2)
            assert b[1 - d] is None and torch.equal(a[d], b[d].cpu()), (label, d)


def test_engine_keygen_encrypt_decrypt_with_hip_samplers():
    """End to end on the GPU with the real sampler: keys, encryption, cc_mult, rotation decode correctly."""
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.csprng import Csprng
    me = ckks_engine(devices=["cuda:0"], logN=12, num_scales=5, num_special_primes=2, is_secured=False)
    assert type(me.rng) is Csprng
    sk = me.create_secret_key()
    pk = me.create_public_key(sk)
    evk = me.create_evk(sk)
    rotk = me.create_rotation_key(sk, 3)
    np.random.seed(2)
    m1, m2 = me.example(-1, 1), me.example(-1, 1)
    c1, c2 = me.encorypt(m1, pk), me.encorypt(m2, pk)
    assert np.abs(me.decrode(c1, sk) - m1).max() < 1e-8
    assert np.abs(me.decrode(me.cc_mult(c1, c2, evk), sk) - m1 * m2).max() < 1e-7
    assert np.abs(me.decrode(me.rotate_single(c1, rotk), sk) - np.roll(m1, 3)).max() < 1e-8


# ---- the three post-processing maps against their EXACT-INTEGER definitions at the gold key shape ------------------
# (they exist in the reference only as CUDA; the oracle restates them, and here the HIP kernels are held directly to the
# mathematical definitions — Python integers — on the tensor shapes gold key generation draws: 43 channels x N = 65536)
GOLD_N = 65536


def _gold_moduli():
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    ctx = ckks_context(logN=16, num_special_primes=4)
    q = [int(x) for x in ctx.q]
    return q + q[-4:]            # 39 own channels + 4 repeating ones = the 43 limbs of a gold uniform draw


def test_randint_gold_shape_equals_exact_128_bit_product():
    from liberate_fhe_amd.csprng import randint_cuda, chacha20_cuda
    q = _gold_moduli()
    C, L = len(q), GOLD_N // 4
    s = random_states(C * L, 77).reshape(C, L, 16)
    qa = np.array(q, dtype=np.uint64)
    d = dev(s)
    blocks = chacha20_cuda.chacha20([dev(s).view(-1, 16)], 1)[0].view(C, L, 16).cpu().numpy()   # the words the fused kernel maps
    got = randint_cuda.randint_fast([d], [qa], -1, 1)[0].cpu().numpy()
    assert got.shape == (C, GOLD_N)
    g = np.random.default_rng(5)
    picks = [(int(c), int(i)) for c, i in zip(g.integers(0, C, 30000), g.integers(0, GOLD_N, 30000))]
    picks += [(c, i) for c in (0, C - 1) for i in (0, 1, 2, 3, GOLD_N - 1)]
    for c, i in picks:
        w = blocks[c, i // 4, 4 * (i % 4):4 * (i % 4) + 4]
        want = co.scale128_exact(q[c], *w) - 1
        assert int(got[c, i]) == want, (c, i)
    assert got.min() >= -1 and all(int(got[c].max()) <= q[c] - 2 for c in range(C))


def test_discrete_gaussian_gold_shape_equals_table_inversion():
    from liberate_fhe_amd.csprng import discrete_gaussian_cuda, chacha20_cuda
    from liberate_fhe_amd.csprng.discrete_gaussian_sampler import cumulative_table
    flat, size, depth = tree()
    table, _ = cumulative_table()
    n = GOLD_N // 4 * 2                       # two polynomials (encrypt's e0, e1)
    s = random_states(n, 78)
    blocks = chacha20_cuda.chacha20([dev(s)], 1)[0].cpu().numpy()
    got = discrete_gaussian_cuda.discrete_gaussian_fast([dev(s)], flat.__array_interface__["data"][0], size, depth, 1)[0].cpu().numpy()
    assert got.shape == (4 * n,)
    idx = np.random.default_rng(6).integers(0, 4 * n, 40000)
    for i in idx:
        w = blocks[i // 4, 4 * (i % 4):4 * (i % 4) + 4]
        assert int(got[i]) == co.cdt_sample_exact(table, *w), int(i)
    assert abs(float(got.astype(np.float64).std()) - 3.2) < 0.05


def test_randround_gold_shape_equals_exact_rule():
    """sign(c) * (floor|c| + [r < rn(frac|c| * 2^32)]) with exact rational arithmetic on the doubles."""
    from fractions import Fraction
    from liberate_fhe_amd.csprng import randround_cuda
    g = np.random.default_rng(9)
    c = g.normal(0.0, 2.0 ** 20, GOLD_N) + g.integers(-3, 4, GOLD_N)
    c[:6] = [0.0, -0.0, 0.5, -0.5, 1.0 - 2.0 ** -33, -(2.0 ** 40 + 0.25)]
    r = g.integers(0, 1 << 32, size=GOLD_N, dtype=np.int64)
    d = dev(r)
    randround_cuda.randround([dev(c)], [d])
    got = d.cpu().numpy()
    for i in list(range(64)) + [int(x) for x in g.integers(0, GOLD_N, 20000)]:
        a = abs(Fraction(float(c[i])))
        ip = a.numerator // a.denominator
        scaled = (a - ip) * (1 << 32)                    # exact; round half to even like __double2ll_rn
        fl = scaled.numerator // scaled.denominator
        rem = scaled - fl
        ifrac = fl + (1 if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1) else 0)
        mag = ip + (1 if int(r[i]) < ifrac else 0)
        want = -mag if np.signbit(c[i]) else mag
        assert int(got[i]) == want, i
