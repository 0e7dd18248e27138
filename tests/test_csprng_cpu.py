"""CPU tests of the sampler path: the oracle against its anchors (RFC 7539 known answer, golden blocks
and CDT table from the reference's own Python code, exact-integer definitions), and the product's
host logic (stream layout, rank invariants) over the checker's kernels against streams recorded from
the REFERENCE's Csprng class."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import csprng_oracle as co
from tests import csprng_streams as cs
from tests.golden import refdriver as rd
from tests.oracle_csprng import oracle_csprng_class

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "csprng_golden.json")))

RFC7539_IN = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, 0x03020100, 0x07060504, 0x0B0A0908, 0x0F0E0D0C,
              0x13121110, 0x17161514, 0x1B1A1918, 0x1F1E1D1C, 0x00000001, 0x09000000, 0x4A000000, 0x00000000]
RFC7539_OUT = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
               0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


def flat_tree():
    bt = np.array([[int(lo), int(hi)] for lo, hi in GOLD["cdt"]["btree_low_high"]], dtype=np.uint64)
    return np.ascontiguousarray(bt.T.ravel()), GOLD["cdt"]["size"], GOLD["cdt"]["depth"]


def test_chacha20_block_rfc7539_known_answer():
    out = co.chacha20_block(np.array([RFC7539_IN], dtype=np.int64))
    assert out[0].tolist() == RFC7539_OUT


def test_chacha20_block_and_counter_match_reference_python():
    g = GOLD["chacha20"]
    states = np.array(g["states"], dtype=np.int64)
    assert np.array_equal(co.chacha20_block(states), np.array(g["blocks"], dtype=np.int64))
    co.step_counter(states, g["step"])
    assert np.array_equal(states, np.array(g["stepped"], dtype=np.int64))


def test_cdt_builder_matches_reference_table():
    from liberate_fhe_amd.csprng.discrete_gaussian_sampler import build_CDT_binary_search_tree, cumulative_table
    btree, ptr, size, depth = build_CDT_binary_search_tree()
    assert (size, depth) == (GOLD["cdt"]["size"], GOLD["cdt"]["depth"]) == (31, 5)
    assert [[str(int(a)), str(int(b))] for a, b in btree] == GOLD["cdt"]["btree_low_high"]
    table, _ = cumulative_table()
    assert table[0] == 0 and all(x < y for x, y in zip(table, table[1:])) and abs(table[-1] - (1 << 127)) < (1 << 70)


def test_scale128_restatement_equals_exact_product():
    g = np.random.default_rng(5)
    w = g.integers(0, 1 << 32, size=(4, 4000), dtype=np.uint64)
    w[:, 0] = 0
    w[:, 1] = 0xFFFFFFFF
    for p in (2, 3, 97, (1 << 32) - 1, 1 << 32, 1099511627689, (1 << 61) - 1, 1152921504606846883, (1 << 64) - 1):
        got = co.scale128(np.uint64(p), *w)
        want = [co.scale128_exact(p, *w[:, i]) for i in range(w.shape[1])]
        assert got.tolist() == want
        assert int(got.max()) < p


def test_cdt_walk_equals_table_inversion():
    from liberate_fhe_amd.csprng.discrete_gaussian_sampler import cumulative_table
    tree, size, depth = flat_tree()
    table, _ = cumulative_table()
    g = np.random.default_rng(6)
    w = g.integers(0, 1 << 32, size=(4, 3000), dtype=np.uint64)
    # land exactly on, just below and just above a few table entries
    for k, i in enumerate((1, 7, 16, 31)):
        for j, delta in enumerate((-1, 0, 1)):
            v = table[i] + delta
            hi, lo = v >> 64, v & ((1 << 64) - 1)
            col = 3 * k + j
            w[0, col], w[1, col] = lo >> 32, lo & 0xFFFFFFFF
            hi2 = (hi << 1) | (col & 1)
            w[2, col], w[3, col] = hi2 >> 32, hi2 & 0xFFFFFFFF
    got = co.cdt_walk(tree, size, depth, *w)
    want = [co.cdt_sample_exact(table, *w[:, i]) for i in range(w.shape[1])]
    assert got.tolist() == want
    assert np.abs(got).max() <= 31


def test_gaussian_statistics():
    tree, size, depth = flat_tree()
    states = np.zeros((1 << 14, 16), dtype=np.int64)
    states[:, :4] = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    states[:, 4:12] = np.arange(8) + 1
    states[:, 12] = np.arange(states.shape[0])
    x = co.discrete_gaussian_fast(states, tree, size, depth, step=states.shape[0]).astype(np.float64)
    assert abs(x.mean()) < 0.06 and abs(x.std() - 3.2) < 0.05
    assert states[0, 12] == states.shape[0]


def test_randround_restatement():
    c = np.array([0.0, -0.0, 0.5, -0.5, 1.25, -1.25, 3.0, -7.0, 2.0 ** 40 + 0.75])
    lo = co.randround(c, np.zeros(len(c), dtype=np.int64))                      # r = 0 < ifrac whenever frac > 0
    hi = co.randround(c, np.full(len(c), 0xFFFFFFFF, dtype=np.int64))           # never rounds up
    assert lo.tolist() == [0, 0, 1, -1, 2, -2, 3, -7, 2 ** 40 + 1]
    assert hi.tolist() == [0, 0, 0, 0, 1, -1, 3, -7, 2 ** 40]
    tie = co.randround(np.array([0.5, 0.5]), np.array([0x7FFFFFFF, 0x80000000], dtype=np.int64))
    assert tie.tolist() == [1, 0]


# ---- host logic of the product Csprng over the checker's kernels --------------------------------------

def replay(r, cfg):
    out = []
    for label, method, kw in cs.stream_calls(cfg):
        res = [t for t in getattr(r, method)(**kw) if t is not None]
        out.append((label, res))
    rr = r.randround(torch.from_numpy(cs.randround_input(cfg["num_coefs"])).to(r.devices[r.local_ids[0]]))
    out.append(("randround", [rr]))
    out.append(("final_states", [s for s in r.states if s is not None]))
    return out


@pytest.mark.parametrize("name", ["small", "gold_like"])
def test_streams_match_reference_class(name):
    """Expected digests were recorded from the reference's Csprng class (make_golden_csprng.py)."""
    entry = GOLD["streams"]["configs"][name]
    cfg = entry["config"]
    r = oracle_csprng_class()(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"],
                              devices=["cpu"] * cfg["n_dev"], seed=GOLD["streams"]["key"], nonce=GOLD["streams"]["nonce"],
                              reference_counter_layout=True)
    for (label, res), want in zip(replay(r, cfg), entry["calls"]):
        assert label == want["label"]
        assert cs.digest([t.cpu() for t in res]) == want["sha256"], label


def test_ranks_reproduce_the_single_process_stream():
    cfg = cs.STREAM_CONFIGS["small"]
    cls = oracle_csprng_class()
    mk = lambda ids: cls(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"],
                         devices=["cpu"] * cfg["n_dev"], seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE, local_ids=ids)
    whole, ranks = mk(None), [mk([d]) for d in range(cfg["n_dev"])]
    rep = cfg["num_repeating_channels"]
    for label, method, kw in cs.stream_calls(cfg):
        a = getattr(whole, method)(**kw)
        for d, r in enumerate(ranks):
            b = getattr(r, method)(**kw)
            assert [x is None for x in b] == [i != d for i in range(cfg["n_dev"])]
            assert torch.equal(a[d], b[d]), (label, d)
        n_rep = kw.get("repeats", 0)
        if n_rep and a[0].dim() == 2:
            assert torch.equal(a[0][-n_rep:], a[1][-n_rep:]), label       # replicated rows agree across GPUs
    c = torch.from_numpy(cs.randround_input(cfg["num_coefs"]))
    r0, r1 = ranks[0].randround(c.clone()), ranks[1].randround(c.clone())
    assert torch.equal(r0, r1)                                             # every rank rounds alike


def test_own_channels_never_share_counters():
    """Default numbering: every state of every GPU has its own counter (the reference's numbering repeats
    GPU 1's counters on GPUs 2, 3, ... — csprng.py:96 — which `reference_counter_layout=True` reproduces)."""
    cfg = cs.STREAM_CONFIGS["gold_like"]
    args = (1024, cfg["num_channels"], cfg["num_repeating_channels"])
    kw = dict(devices=["cpu"] * cfg["n_dev"], seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE)
    cls = oracle_csprng_class()
    own = lambda r: torch.cat([r.counters[d][: r.shares[d] * r.L] for d in range(r.num_devices)])
    mine, like_ref = cls(*args, **kw), cls(*args, reference_counter_layout=True, **kw)
    assert own(mine).unique().numel() == own(mine).numel() == sum(cfg["num_channels"]) * 256
    assert own(mine).max() < mine.repeating_start
    assert own(like_ref).unique().numel() < own(like_ref).numel()


def test_argument_errors():
    cls = oracle_csprng_class()
    with pytest.raises(Exception):
        cls(64, [1, 2, 3], 1, devices=["cpu"] * 2)
    with pytest.raises(ValueError):
        cls(64, [2], 1, devices=["cpu"], seed=[1, 2, 3])
    r = cls(64, [2], 1, devices=["cpu"], seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE)
    with pytest.raises(ValueError):
        r.randint([[5, 7, 11, 13]], repeats=1)          # 3 own channels requested, 2 procured
    with pytest.raises(ValueError):
        r.discrete_gaussian(repeats=2)


def test_product_samplers_refuse_cpu_tensors():
    """No CPU fallback behind the product class: the HIP shims reject host tensors."""
    from liberate_fhe_amd.csprng import Csprng
    r = Csprng(64, [1], 1, devices=["cpu"], seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE)
    with pytest.raises(RuntimeError):
        r.randint(amax=3, shift=-1, repeats=1)
    with pytest.raises(RuntimeError):
        r.discrete_gaussian(repeats=1)


@pytest.mark.reference
@pytest.mark.skipif(not rd.reference_available(), reason="needs /root/reference")
def test_host_logic_equals_reference_class_live():
    cfg = cs.STREAM_CONFIGS["small"]
    ref = rd.reference_csprng(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"], cfg["n_dev"],
                              cs.STREAM_KEY, cs.STREAM_NONCE)
    me = oracle_csprng_class()(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"],
                               devices=["cpu"] * cfg["n_dev"], seed=cs.STREAM_KEY, nonce=cs.STREAM_NONCE,
                               reference_counter_layout=True)
    assert me.inc == ref.inc and me.L == ref.L and me.shares == ref.shares
    for a, b in zip(ref.states, me.states):
        assert torch.equal(a, b)
    for label, method, kw in cs.stream_calls(cfg):
        for a, b in zip(getattr(ref, method)(**kw), getattr(me, method)(**kw)):
            assert torch.equal(a, b), label
