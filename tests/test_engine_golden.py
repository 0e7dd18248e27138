"""Engine hot path against the golden digests captured from the reference engine
(tests/golden/engine_digests.json, generator tests/golden/make_golden.py).

CPU leg: orchestration + checker backend (pins the composition-level oracle to the reference).
GPU leg: the product — HIP kernels through the C ABI — on the same seeded inputs, bit-exact.
"""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from liberate_fhe_amd.utils import synth

warnings.filterwarnings("ignore", category=UserWarning)
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "engine_digests.json")))


def digest(ct):
    out = []
    for comp in ct.data:
        h = hashlib.sha256()
        for t in comp:
            h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
        out.append({"sha256": h.hexdigest(), "shape": [list(t.shape) for t in comp],
                    "head": [int(x) for x in comp[0][0, :4].cpu()], "tail": [int(x) for x in comp[0][-1, -4:].cpu()]})
    return out


def check_config(engine, rec):
    assert [int(x) for x in engine.ctx.q] == rec["q"] and engine.hash == rec["hash"]
    s = rec["seeds"]
    a, b = synth.ciphertext(engine, s["ct_a"], 0), synth.ciphertext(engine, s["ct_b"], 0)
    evk = synth.key_switch_key(engine, s["evk"])
    rotk = synth.key_switch_key(engine, s["rotk"], origin=f"rotation key:{s['rot_delta']}")
    ops = rec["ops"]
    assert digest(engine.rescale(a)) == ops["rescale(a)"]
    prod = engine.cc_mult(a, b, evk)
    assert digest(prod) == ops["cc_mult(a,b,evk)"]
    assert digest(engine.rotate_single(a, rotk)) == ops["rotate_single(a,rotk)"]
    assert digest(engine.rotate_single(prod, rotk)) == ops["rotate_single(cc_mult,rotk)"]
    assert digest(engine.cc_add(a, b)) == ops["cc_add(a,b)"]
    if "cc_mult(prod,prod,evk)" in ops:
        assert digest(engine.cc_mult(prod, prod, evk)) == ops["cc_mult(prod,prod,evk)"]


@pytest.mark.parametrize("name", ["small", "small_x2", "bronze", "silver", "sb30", "sb45"])
def test_checker_engine_reproduces_reference_digests(name):
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    rec = GOLD[name]
    eng = ckks_engine(devices=["cpu"] * rec["n_devices"], backend=OracleBackend(), **rec["params"])
    check_config(eng, rec)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "small_x2", "bronze", "silver", "gold", "gold_x8", "platinum", "sb30", "sb45"])
def test_hip_engine_reproduces_reference_digests(name):
    """All four presets (platinum: logN 17, 6 special primes, a five-stage strided pass — the LDS-tiled fallback
    of the column kernels) and the other scale-prime widths: sb30 = 30-bit scale primes (fp64 class), sb45 =
    45-bit scale primes (>= 2^41: the integer class for every limb)."""
    from liberate_fhe_amd.fhe import ckks_engine
    rec = GOLD[name]
    eng = ckks_engine(devices=["cuda:0"] * rec["n_devices"], **rec["params"])
    check_config(eng, rec)


def check_levels(engine, rec):
    assert [int(x) for x in engine.ctx.q] == rec["q"] and engine.hash == rec["hash"]
    s = rec["seeds"]
    evk = synth.key_switch_key(engine, s["evk"])
    rotk = synth.key_switch_key(engine, s["rotk"], origin=f"rotation key:{s['rot_delta']}")
    for level in rec["levels"]:
        a, b = synth.ciphertext(engine, s["ct_a"], level), synth.ciphertext(engine, s["ct_b"], level)
        assert digest(engine.cc_mult(a, b, evk)) == rec["ops"][f"cc_mult(a,b,evk)@{level}"], level
        assert digest(engine.rotate_single(a, rotk)) == rec["ops"][f"rotate_single(a,rotk)@{level}"], level
        assert digest(engine.rescale(a)) == rec["ops"][f"rescale(a)@{level}"], level


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gold_levels", "gold_levels_x8"])
def test_hip_engine_gold_below_level_zero(name):
    """Gold at levels 10 / 20 / 32 against the reference engine, on one device and on 8 logical devices: rows per device
    shrink with the level and whole devices run out of rows (7 alive at level 10, 4 at 20, 1 at 32 and 33) —
    rns_partition.py:64-170, ckks_engine.py:746-904 (`len_devices`, parts of an exhausted device)."""
    from liberate_fhe_amd.fhe import ckks_engine
    rec = GOLD[name]
    eng = ckks_engine(devices=["cuda:0"] * rec["n_devices"], **rec["params"])
    if rec["n_devices"] == 8:
        assert [eng.len_devices[l] for l in (0, 10, 20, 32)] == [8, 7, 4, 1]
    check_levels(eng, rec)


def natural_rows(eng, ct):
    """Components of a ciphertext as [rows, N] arrays with the rows in the order of the prime chain."""
    dest = eng.ntt.p.destination_arrays[ct.level]
    out = []
    for comp in ct.data:
        rows = {}
        for d, t in enumerate(comp):
            arr = t.cpu().numpy()
            for r, prime in enumerate(dest[d]):
                rows[prime] = arr[r]
        out.append(np.stack([rows[k] for k in sorted(rows)]))
    return out


@pytest.mark.gpu
def test_hip_gold_balanced_limb_map_equals_one_device_row_by_row():
    """ckks_engine(balanced_limb_map=True) = rns_partition(35, 4, 8, balance=True): the top digit on device 1 instead of device 0
    (rows 5 / 6 / 4 x 6 instead of 7 / 4 x 7; NOT the reference's layout, default off).  Digits are unchanged, so every canonical
    word per limb is: cc_mult / rotate at level 0, across the rescale 3 -> 4 (device 0 loses its scale digit and keeps the base
    prime), 9 -> 10 (device 7 runs out of rows) and at level 31 (two devices left) equal the undivided engine in prime order."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    outs = []
    for n_dev, bal in ((1, False), (8, True)):
        eng = ckks_engine(devices=["cuda:0"] * n_dev, balanced_limb_map=bal, **params)
        if bal:
            assert [len(r) for r in eng.ntt.p.destination_arrays[0]] == [5, 6, 4, 4, 4, 4, 4, 4]
            assert eng.len_devices[10] == 7 and eng.len_devices[31] == 2
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        res = []
        for lvl in (0, 3, 9, 30):
            a, b = synth.ciphertext(eng, 3 + lvl, lvl), synth.ciphertext(eng, 4 + lvl, lvl)
            res += [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
            res.append(eng.rotate_single(res[-2], rotk))
        outs.append([natural_rows(eng, ct) for ct in res])
        del eng, evk, rotk
        torch.cuda.empty_cache()
    for one, eight in zip(*outs):
        for x, y in zip(one, eight):
            assert x.shape == y.shape and (x == y).all()


@pytest.mark.gpu
def test_balanced_limb_map_decrypts_at_every_level():
    """The reference's final scaling reads a scale-prime row beside the base prime on device 0 (eng.py:517-533).  In the balanced
    map device 0 keeps one only while its own scale digit lives (small over 2 devices: levels 0 and 1); from level 2 on the
    scaler is the first row of device 1, computed there and moved: the same plaintext as the reference layout at every level,
    also after a multiplication that crosses the boundary."""
    from liberate_fhe_amd.fhe import ckks_engine
    params = GOLD["small"]["params"]
    ref = ckks_engine(devices=["cuda:0"] * 2, **params)
    bal = ckks_engine(devices=["cuda:0"] * 2, balanced_limb_map=True, **params)
    assert [len(r) for r in ref.ntt.p.destination_arrays[0]] == [4, 2] and [len(r) for r in bal.ntt.p.destination_arrays[0]] == [3, 3]
    assert [len(r) for r in bal.ntt.p.destination_arrays[2]][0] == 1 and bal._scaler_dev[:4] == [0, 0, 1, 1] and not any(ref._scaler_dev)
    sk = bal.create_secret_key()
    pk, evk = bal.create_public_key(sk), bal.create_evk(sk)
    np.random.seed(3)
    m = bal.example(-1, 1)
    for level in range(bal.num_levels - 1):
        ct = bal.encorypt(m, pk, level=level)
        assert np.abs(bal.decrode(ct, sk) - m).max() < 1e-5, level
    prod = bal.cc_mult(bal.encorypt(m, pk, level=1), bal.encorypt(m, pk, level=1), evk)      # level 1 -> 2
    assert prod.level == 2 and np.abs(bal.decrode(prod, sk) - m * m).max() < 1e-4
    # evaluation words equal the reference layout's, limb by limb
    rotk_b, rotk_r = synth.key_switch_key(bal, 6, origin="rotation key:1"), synth.key_switch_key(ref, 6, origin="rotation key:1")
    a_b, a_r = synth.ciphertext(bal, 21, 2), synth.ciphertext(ref, 21, 2)
    for x, y in zip(natural_rows(bal, bal.rotate_single(a_b, rotk_b)), natural_rows(ref, ref.rotate_single(a_r, rotk_r))):
        assert (x == y).all()


@pytest.mark.gpu
def test_ops_of_one_engine_alternating_between_two_streams_are_ordered_not_raced():
    """The engine's scratch (digits, sums, operand stack, plan workspaces) is per engine and lane, not per stream.  An op that
    arrives on another current stream than the lane's previous op makes its stream wait for the previous one (_same_stream): 40
    ops alternating between two streams with no synchronisation in between give the single-stream words."""
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **GOLD["silver"]["params"])
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:1")
    want_m, want_r = eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)
    want_b = eng.rotate_single_batch([a, b, a, b, a], rotk)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    for i in range(20):
        with torch.cuda.stream(s1):
            got.append(("m", eng.cc_mult(a, b, evk)))
        with torch.cuda.stream(s2):
            got.append(("r", eng.rotate_single(a, rotk)))
        if i % 7 == 3:
            with torch.cuda.stream(s1):
                got.append(("b", eng.rotate_single_batch([a, b, a, b, a], rotk)))
    torch.cuda.synchronize()
    for kind, ct in got:
        if kind == "b":
            for x, y in zip(ct, want_b):
                assert torch.equal(x.data[0][0], y.data[0][0]) and torch.equal(x.data[1][0], y.data[1][0])
            continue
        w = want_m if kind == "m" else want_r
        assert torch.equal(ct.data[0][0], w.data[0][0]) and torch.equal(ct.data[1][0], w.data[1][0]), kind


@pytest.mark.gpu
def test_compact_key_halves_a_keys_memory_and_changes_no_result():
    """engine.compact_key(): the raw pack of a key the engine made is freed (the fused key switch reads the planes copy only);
    expand_key() restores it from the planes — integer-class rows byte for byte, fp64-class rows as the canonical residues of the
    lazy words they held.  Same cc_mult / rotate words before, while compact, and after."""
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **GOLD["silver"]["params"])
    sk = eng.create_secret_key()
    evk = eng.create_evk(sk)
    rotk = eng.create_rotation_key(sk, 1)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    want = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
    raw = [part.data[c][0].clone() for part in evk.data for c in range(2)]
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    freed = eng.compact_key(evk)
    pack_bytes = len(evk.data) * 2 * evk.data[0].data[0][0].size(0) * eng.ctx.N * 8
    assert freed == pack_bytes and eng.compact_key(evk) == 0
    assert before - torch.cuda.memory_allocated() >= freed - (4 << 20)
    for _ in range(2):
        got = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
        for g, w in zip(got, want):
            assert torch.equal(g.data[0][0], w.data[0][0]) and torch.equal(g.data[1][0], w.data[1][0])
    eng.expand_key(evk)
    q = [eng.ctx.q[i] for i in eng.ntt.p.d_special[0]]
    k = 0
    for part in evk.data:
        for c in range(2):
            now, old = part.data[c][0], raw[k]
            k += 1
            for r, qr in enumerate(q):
                if qr >= (1 << 41):
                    assert torch.equal(now[r], old[r])
                else:
                    assert bool(((now[r] >= 0) & (now[r] < qr)).all()) and bool((((old[r] - now[r]) % qr) == 0).all())
    got = eng.cc_mult(a, b, evk)
    assert torch.equal(got.data[0][0], want[0].data[0][0]) and torch.equal(got.data[1][0], want[0].data[1][0])
    eng.compact_key(evk)
    eng.release_key(evk)                              # the planes are a compact key's only copy: the raw words come back first
    got = eng.cc_mult(a, b, evk)
    assert torch.equal(got.data[0][0], want[0].data[0][0])
    # a Galois key set: every rotation key of it
    galk = eng._new([eng.create_rotation_key(sk, d) for d in (1, 2)], "galois key", include_special=True, ntt_state=True, montgomery_state=True)
    r_before = eng.rotate_single(a, galk.data[1])
    assert eng.compact_key(galk) == 2 * pack_bytes
    assert torch.equal(eng.rotate_single(a, galk.data[1]).data[1][0], r_before.data[1][0])
    eng.expand_key(galk)
    assert torch.equal(eng.rotate_single(a, galk.data[1]).data[1][0], r_before.data[1][0])
    # a key whose tensors are the caller's (here: clones of the parts) is never touched
    foreign = evk._replace(data=[part._replace(data=([t.clone() for t in part.data[0]], [t.clone() for t in part.data[1]])) for part in evk.data])
    got = eng.cc_mult(a, b, foreign)
    assert torch.equal(got.data[0][0], want[0].data[0][0])
    with pytest.raises(ValueError):
        eng.compact_key(foreign)


@pytest.mark.gpu
def test_hip_gold_eight_logical_devices_equal_one_device():
    """BASELINE configs[3]'s partition (rns_partition(35, 4, 8): 11 / 8 / .. / 8 rows with the special limbs, digit
    exchange between 8 shards) against the undivided engine, row by row in prime order, at level 0 and across the
    rescale 9 -> 10 that leaves device 7 without rows."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    outs = []
    for n_dev in (1, 8):
        eng = ckks_engine(devices=["cuda:0"] * n_dev, **params)
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        a9, b9 = synth.ciphertext(eng, 7, 9), synth.ciphertext(eng, 8, 9)
        res = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk), eng.cc_mult(a9, b9, evk)]
        res.append(eng.rotate_single(res[-1], rotk))
        outs.append([natural_rows(eng, ct) for ct in res])
        del eng, evk, rotk
        torch.cuda.empty_cache()
    for one, eight in zip(*outs):
        for x, y in zip(one, eight):
            assert x.shape == y.shape and (x == y).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "silver"])
def test_hip_public_relinearize_of_an_exact_triplet(name):
    """cc_mult(relin=False) hands out the reference's exact NTT / Montgomery / lazy triplet (not the internal
    plain-domain one); relinearize() of it must land on the same ciphertext as the fused cc_mult — the reference
    engine's digest."""
    from liberate_fhe_amd.fhe import ckks_engine
    rec = GOLD[name]
    eng = ckks_engine(devices=["cuda:0"], **rec["params"])
    s = rec["seeds"]
    a, b = synth.ciphertext(eng, s["ct_a"], 0), synth.ciphertext(eng, s["ct_b"], 0)
    evk = synth.key_switch_key(eng, s["evk"])
    trip = eng.cc_mult(a, b, evk, relin=False)
    assert trip.origin == "cipher text triplet" and trip.ntt_state and trip.montgomery_state
    assert digest(eng.relinearize(trip, evk)) == rec["ops"]["cc_mult(a,b,evk)"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_dev", [("small", 1), ("small", 3), ("silver", 1), ("silver", 2)])
def test_hip_engine_equals_checker_engine_on_fresh_seeds(name, n_dev):
    """Seeds and device counts the fixtures do not cover: HIP engine vs the oracle composition."""
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    params = GOLD[name]["params"]
    hip = ckks_engine(devices=["cuda:0"] * n_dev, **params)
    chk = ckks_engine(devices=["cpu"] * n_dev, backend=OracleBackend(), **params)
    outs = []
    for eng in (hip, chk):
        a, b = synth.ciphertext(eng, 101, 0), synth.ciphertext(eng, 102, 0)
        evk, rotk = synth.key_switch_key(eng, 103), synth.key_switch_key(eng, 104, origin="rotation key:129")
        conjk = synth.key_switch_key(eng, 105, origin="conjugation key")
        prod = eng.cc_mult(a, b, evk)
        deep = eng.cc_mult(prod, a if False else eng.level_up(a, 1), evk)
        outs.append([digest(x) for x in (prod, deep, eng.rotate_single(deep, rotk), eng.conjugate(a, conjk),
                                         eng.rescale(a), eng.cc_sub(a, b), eng.cc_mult(a, b, evk, relin=False))])
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_hip_engine_end_to_end_decode_error():
    """configs[2]: silver cc_mult + relinearize with real keys; decoded product within the reference's
    own accuracy (~5e-8 on +-1 data) and identical across device splits."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    eng = ckks_engine(**{**presets.params["silver"], "devices": ["cuda:0"]})
    sk = eng.create_secret_key()
    pk = eng.create_public_key(sk)
    evk = eng.create_evk(sk)
    rotk = eng.create_rotation_key(sk, 11)
    np.random.seed(3)
    m1, m2 = eng.example(-1, 1), eng.example(-1, 1)
    c1, c2 = eng.encorypt(m1, pk), eng.encorypt(m2, pk)
    assert np.abs(eng.decrode(c1, sk) - m1).max() < 1e-8
    prod = eng.cc_mult(c1, c2, evk)
    assert np.abs(eng.decrode(prod, sk) - m1 * m2).max() < 2e-7
    assert np.abs(eng.decrode(eng.rotate_single(prod, rotk), sk) - np.roll(m1 * m2, 11)).max() < 2e-7
    assert np.abs(eng.decrode(eng.cc_add(c1, c2), sk) - (m1 + m2)).max() < 1e-8


def _rot_params():
    # a small two-pass ring (the fused key-switch core needs logN >= 13) with two digits
    return dict(logN=13, num_scales=5, num_special_primes=2, is_secured=False)


def test_rotate_single_batch_checker_equals_loop():
    """Host logic of rotate_single_batch (grouping 4 / 2 / 1, mixed levels, fallbacks) through the checker backend."""
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), **_rot_params())
    rotk = synth.key_switch_key(eng, 7, origin="rotation key:3")
    cts = [synth.ciphertext(eng, 200 + i, 0) for i in range(3)] + [synth.ciphertext(eng, 300, 1)]
    want = [digest(eng.rotate_single(ct, rotk)) for ct in cts]
    got = [digest(x) for x in eng.rotate_single_batch(cts, rotk)]
    assert got == want
    # edge cases: nothing to do, one ciphertext, a wrong key type
    assert eng.rotate_single_batch([], rotk) == []
    assert [digest(x) for x in eng.rotate_single_batch(cts[:1], rotk)] == want[:1]
    from liberate_fhe_amd.fhe.presets import errors
    with pytest.raises(errors.NotMatchType):
        eng.rotate_single_batch(cts, synth.key_switch_key(eng, 8))
    assert eng.cc_mult_batch([], synth.key_switch_key(eng, 8)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("name,count", [("small13", 7), ("silver", 6), ("gold", 4)])
def test_hip_rotate_single_batch_equals_loop(name, count):
    """configs[4] (rotate batched): groups of 4 / 2 ciphertexts through lf_ks_core_batch give, bit for bit, what
    one rotate_single per ciphertext gives; for the small ring also what the checker composition gives."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    params = _rot_params() if name == "small13" else dict(presets.params[name])
    params.pop("devices", None)
    eng = ckks_engine(devices=["cuda:0"], **params)
    rotk = synth.key_switch_key(eng, 7, origin="rotation key:3")
    cts = [synth.ciphertext(eng, 200 + i, 0) for i in range(count - 1)] + [synth.ciphertext(eng, 300, 1)]
    want = [digest(eng.rotate_single(ct, rotk)) for ct in cts]
    got = [digest(x) for x in eng.rotate_single_batch(cts, rotk)]
    assert got == want
    if name == "small13":
        from tests.oracle_backend import OracleBackend
        chk = ckks_engine(devices=["cpu"], backend=OracleBackend(), **params)
        rk = synth.key_switch_key(chk, 7, origin="rotation key:3")
        cc = [synth.ciphertext(chk, 200 + i, 0) for i in range(count - 1)] + [synth.ciphertext(chk, 300, 1)]
        assert [digest(chk.rotate_single(ct, rk)) for ct in cc] == want


def test_cc_mult_batch_checker_equals_loop():
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), **_rot_params())
    evk = synth.key_switch_key(eng, 9)
    pairs = [(synth.ciphertext(eng, 400 + i, 0), synth.ciphertext(eng, 500 + i, 0)) for i in range(3)]
    want = [digest(eng.cc_mult(a, b, evk)) for a, b in pairs]
    assert [digest(x) for x in eng.cc_mult_batch(pairs, evk)] == want


@pytest.mark.gpu
@pytest.mark.parametrize("name,count", [("small13", 7), ("silver", 6), ("gold", 4)])
def test_hip_cc_mult_batch_equals_loop(name, count):
    """Groups of 4 / 2 multiplications under one evaluation key (shared inverse-transform launch + lf_ks_core_batch)
    equal one cc_mult per pair, bit for bit; mixed levels fall into separate groups."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    params = _rot_params() if name == "small13" else dict(presets.params[name])
    params.pop("devices", None)
    eng = ckks_engine(devices=["cuda:0"], **params)
    evk = synth.key_switch_key(eng, 9)
    pairs = [(synth.ciphertext(eng, 400 + i, 0), synth.ciphertext(eng, 500 + i, 0)) for i in range(count - 1)]
    pairs.append((synth.ciphertext(eng, 600, 1), synth.ciphertext(eng, 601, 1)))
    want = [digest(eng.cc_mult(a, b, evk)) for a, b in pairs]
    assert [digest(x) for x in eng.cc_mult_batch(pairs, evk)] == want


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 2, 3, 5, 6])
def test_moddown_ws_equals_chunked_kernel(K):
    """lf_ks_moddown_ws (pivots once per coefficient + closed form, template on K) against lf_ks_moddown_batch (the
    reference's chain per row chunk, itself pinned by the engine digests), with and without addends / Galois gather."""
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], logN=13, num_scales=6, num_special_primes=K, is_secured=False)
    level, d = 0, 0
    tabs = eng._ks_tables(level)
    N = eng.ctx.N
    rows, ell = eng._rows(d, level, True), eng._rows(d, level, False)
    cs = eng._consts(d, level, True)
    dest = eng.ntt.p.destination_arrays_with_special[level][d]
    rng = np.random.default_rng(50 + K)
    def canon(n_rows, ids):
        return torch.from_numpy(np.stack([rng.integers(0, eng.ctx.q[i], size=N, dtype=np.int64) for i in ids[:n_rows]])).cuda()
    s = [canon(rows, dest) for _ in range(4)]
    adds = [canon(ell, dest), None, canon(ell, dest), canon(ell, dest)]
    rs = eng._vec("Rs", d, level, True)
    for gal in (None, (pow(5, -1, 2 * N), eng._vec("_2q", d, level, False))):
        want = [torch.empty((ell, N), dtype=torch.int64, device="cuda") for _ in range(4)]
        got = [torch.empty((ell, N), dtype=torch.int64, device="cuda") for _ in range(4)]
        eng.backend.ks_moddown_batch(s, want, adds, ell, K, tabs[("pir", d)], rs, cs, PiP=tabs[("pip", d)], galois=gal)
        ws = torch.empty(eng.backend.moddown_ws_words(4, ell, K, N), dtype=torch.int64, device="cuda")
        eng.backend.ks_moddown_ws(s, got, adds, ell, K, ws, tabs[("pir", d)], rs, cs, PiP=tabs[("pip", d)], galois=gal)
        for w, g in zip(want, got):
            assert torch.equal(w, g)


@pytest.mark.gpu
def test_hip_batch_calls_on_a_cold_engine():
    """The very first operations of an engine are batched ones: key pack, per-level tables and the fp64 twiddle twins
    are built by the first group on the caller's stream before the second lane is forked."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    params = dict(presets.params["silver"])
    params.pop("devices", None)
    eng = ckks_engine(devices=["cuda:0"], **params)
    rotk = synth.key_switch_key(eng, 7, origin="rotation key:5")
    evk = synth.key_switch_key(eng, 9)
    cts = [synth.ciphertext(eng, 700 + i, 0) for i in range(12)]
    pairs = [(cts[i], cts[(i + 5) % 12]) for i in range(12)]
    got_r = [digest(x) for x in eng.rotate_single_batch(cts, rotk)]
    got_m = [digest(x) for x in eng.cc_mult_batch(pairs, evk)]
    ref = ckks_engine(devices=["cuda:0"], **params)
    assert got_r == [digest(ref.rotate_single(ct, rotk)) for ct in cts]
    assert got_m == [digest(ref.cc_mult(a, b, evk)) for a, b in pairs]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [3, 8, 13])
def test_hip_engine_equals_checker_engine_at_deeper_levels(level):
    """Fewer live limbs: partial last digit, other part counts, other (ell, K) shapes of the mod-down — HIP vs checker
    for cc_mult, rotate_single and their batched forms at silver levels the fixtures do not reach."""
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    params = GOLD["silver"]["params"]
    hip = ckks_engine(devices=["cuda:0"], **params)
    chk = ckks_engine(devices=["cpu"], backend=OracleBackend(), **params)
    outs = []
    for eng in (hip, chk):
        a, b = synth.ciphertext(eng, 801, level), synth.ciphertext(eng, 802, level)
        evk, rotk = synth.key_switch_key(eng, 803), synth.key_switch_key(eng, 804, origin="rotation key:17")
        res = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
        if eng is hip:
            res += eng.cc_mult_batch([(a, b), (b, a)], evk) + eng.rotate_single_batch([a, b], rotk)
        else:
            res += [eng.cc_mult(a, b, evk), eng.cc_mult(b, a, evk), eng.rotate_single(a, rotk), eng.rotate_single(b, rotk)]
        outs.append([digest(x) for x in res])
    assert outs[0] == outs[1]


@pytest.mark.gpu
@pytest.mark.parametrize("name,splits", [("silver", [(0, 9)]), ("silver", [(0, 1), (1, 5), (6, 3)]), ("gold", [(0, 4), (4, 6)])])
def test_ks_fwd_groups_plus_tail_equal_ks_core(name, splits):
    """lf_ks_fwd over any partition of the digits into consecutive groups, then lf_ks_tail, leaves exactly what the
    undivided lf_ks_core leaves (the limb-sharded engine feeds the groups as they arrive from the other GPUs)."""
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **GOLD[name]["params"])
    level, d = 0, 0
    a = synth.ciphertext(eng, 7, level)
    key = synth.key_switch_key(eng, 8)
    tabs = eng._ks_tables(level)
    N, logN = eng.ctx.N, eng.ctx.logN
    rows, ell = eng._rows(d, level, True), eng._rows(d, level, False)
    nparts = len(tabs["order"])
    assert sum(c for _, c in splits) == nparts
    st = torch.empty((ell, N), dtype=torch.int64, device="cuda:0")
    n_d, desc_d, tab_d = tabs[("digits", d)]
    eng.backend.ks_digits(a.data[1][0], st, n_d, desc_d, tab_d, eng._consts(d, level, False))
    desc, E, Ed = tabs[("extend", d)]
    cs = eng._consts(d, level, True)
    kp = eng._key_pack(key)[0]
    tw, itw, ninv = eng._tw(d, level, True), eng._tw(d, level, True, True), eng._vec("Ninv", d, level, True)
    ext1, ext2 = (torch.zeros((nparts, rows, N), dtype=torch.int64, device="cuda:0") for _ in range(2))
    s1, s2 = (torch.zeros((2, rows, N), dtype=torch.int64, device="cuda:0") for _ in range(2))
    eng.backend.ks_core(st, nparts, rows, logN, desc, E, Ed, kp, tabs["first_part"], eng.ntt.starts[level][d], ext1, s1, tw, itw, ninv, cs)
    for first, count in splits:
        eng.backend.ks_fwd(st, first, count, rows, logN, desc, E, Ed, ext2, tw, cs)
    eng.backend.ks_tail(nparts, rows, logN, kp, tabs["first_part"], eng.ntt.starts[level][d], ext2, s2, itw, ninv, cs)
    assert torch.equal(ext1, ext2) and torch.equal(s1, s2)


@pytest.mark.gpu
def test_a_knob_flipped_between_the_halves_of_an_op_is_refused_not_misread():
    """lf_tune is process-wide; the halves of an op may be separate native calls whose scratch format follows a knob: the extended
    digits between lf_ks_fwd and lf_ks_tail (LF_TUNE_DIGIT_PLANES), cc_mult's operand stack between lf_cc_mult_evk_pre(which = 1),
    its (which = 2) and lf_cc_mult_evk_post (LF_TUNE_MORE_PLANES bit 1), the workspace between lf_ntt_pass_ws 1 and 2
    (LF_TUNE_WS_EXTRA_STAGE).  The producer notes the format it wrote, the consumer gets LF_ERR_STATE (10002) — nothing launched, the
    result buffer untouched — when it would read another one; with the knob back, or the first half repeated, the op completes
    with the undivided call's words."""
    import ctypes
    from liberate_fhe_amd._native import lib, HipError, LF_ERR_STATE
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **GOLD["silver"]["params"])
    level, d = 0, 0
    a, b = synth.ciphertext(eng, 7, level), synth.ciphertext(eng, 9, level)
    key = synth.key_switch_key(eng, 8)
    tabs = eng._ks_tables(level)
    N, logN = eng.ctx.N, eng.ctx.logN
    rows, ell = eng._rows(d, level, True), eng._rows(d, level, False)
    nparts = len(tabs["order"])
    st = torch.empty((ell, N), dtype=torch.int64, device="cuda:0")
    n_d, desc_d, tab_d = tabs[("digits", d)]
    eng.backend.ks_digits(a.data[1][0], st, n_d, desc_d, tab_d, eng._consts(d, level, False))
    desc, E, Ed = tabs[("extend", d)]
    cs = eng._consts(d, level, True)
    kp = eng._key_pack(key)[0]
    tw, itw, ninv = eng._tw(d, level, True), eng._tw(d, level, True, True), eng._vec("Ninv", d, level, True)
    ext1, ext2 = (torch.zeros((nparts, rows, N), dtype=torch.int64, device="cuda:0") for _ in range(2))
    s1 = torch.zeros((2, rows, N), dtype=torch.int64, device="cuda:0")
    s2 = torch.full((2, rows, N), -7, dtype=torch.int64, device="cuda:0")
    eng.backend.ks_core(st, nparts, rows, logN, desc, E, Ed, kp, tabs["first_part"], eng.ntt.starts[level][d], ext1, s1, tw, itw, ninv, cs)
    tail = lambda: eng.backend.ks_tail(nparts, rows, logN, kp, tabs["first_part"], eng.ntt.starts[level][d], ext2, s2, itw, ninv, cs)
    assert lib.lf_tune(3, -1) == 1
    try:
        # 1. digits written as planes, the tail asked to read raw words
        eng.backend.ks_fwd(st, 0, nparts, rows, logN, desc, E, Ed, ext2, tw, cs)
        lib.lf_tune(3, 0)
        with pytest.raises(HipError, match="LF_ERR_STATE"):
            tail()
        torch.cuda.synchronize()
        assert bool((s2 == -7).all()), "the refused half launched something"
        lib.lf_tune(3, 1)
        tail()                                           # knob back: the digits are still good
        assert torch.equal(s1, s2)
        # 2. the other way round, and only PART of the digits rewritten under the new setting
        lib.lf_tune(3, 0)
        eng.backend.ks_fwd(st, 0, nparts, rows, logN, desc, E, Ed, ext2, tw, cs)
        lib.lf_tune(3, 1)
        eng.backend.ks_fwd(st, 0, 2, rows, logN, desc, E, Ed, ext2, tw, cs)
        s2.fill_(-7)
        with pytest.raises(HipError, match="LF_ERR_STATE"):
            tail()
        eng.backend.ks_fwd(st, 2, nparts - 2, rows, logN, desc, E, Ed, ext2, tw, cs)     # the rest as well: consistent again
        tail()
        assert torch.equal(s1, s2)
    finally:
        lib.lf_tune(3, 1)
    # 3. cc_mult in the pieces a sharded rank issues (lf_cc_mult_evk_pre which = 1 | 2, lf_ks_plan_fwd, lf_cc_mult_evk_post): the
    #    operand stack is written by pre(1), transformed by pre(2), read by post
    from liberate_fhe_amd.fhe.backend import _ds
    be = eng.backend
    want = eng.cc_mult(a, b, key)
    plan, _, first_part, row_off = eng._op_plan(1, d)
    polys = [a.data[0][0], a.data[1][0], b.data[0][0], b.data[1][0]]
    rbuf = torch.stack([t[0] for t in polys]).contiguous()                       # the dropped limb's rows
    ins = (ctypes.c_void_p * 4)(*[t.data_ptr() + N * 8 for t in polys])          # the surviving rows start behind it
    row0s = (ctypes.c_void_p * 4)(*[rbuf[k].data_ptr() for k in range(4)])
    stream = _ds(polys[0])[1]
    state = eng._ws("ks_state", (plan.ell, N), d)
    out = torch.full((2, plan.ell, N), -7, dtype=torch.int64, device="cuda:0")
    assert lib.lf_tune(5, -1) == 3 and lib.lf_stack_planes(logN, plan.ell, plan.q_host) == 1
    try:
        be.cc_mult_pre(plan, ins, row0s, stream, which=1)                        # the stack leaves as planes
        lib.lf_tune(5, 1)                                                        # .. and is now wanted as raw words
        assert lib.lf_stack_planes(logN, plan.ell, plan.q_host) == 0
        with pytest.raises(HipError, match="LF_ERR_STATE"):
            be.cc_mult_pre(plan, None, None, stream, which=2)
        lib.lf_tune(5, 3)
        be.cc_mult_pre(plan, None, None, stream, which=2)
        be.plan_fwd(plan, state, 0, plan.nparts, True)
        lib.lf_tune(5, 1)
        with pytest.raises(HipError, match="LF_ERR_STATE"):
            be.cc_mult_post(plan, kp, first_part, row_off, out)
        torch.cuda.synchronize()
        assert bool((out == -7).all()), "the refused half launched something"
        lib.lf_tune(5, 3)
        be.cc_mult_post(plan, kp, first_part, row_off, out)
        assert torch.equal(out[0], want.data[0][0]) and torch.equal(out[1], want.data[1][0])
        # the whole op again with raw stacks (knob 5 = 1: sums as planes, stack raw): same words
        lib.lf_tune(5, 1)
        again = eng.cc_mult(a, b, key)
        assert torch.equal(again.data[0][0], want.data[0][0]) and torch.equal(again.data[1][0], want.data[1][0])
    finally:
        lib.lf_tune(5, 3)


@pytest.mark.gpu
def test_the_split_of_a_workspace_transform_cannot_change_between_its_two_launches():
    """lf_ntt_pass_ws(1) under LF_TUNE_WS_EXTRA_STAGE = 1 leaves 5 + 11 stages' worth in the workspace; the tiled pass launched
    under 0 would run 12: LF_ERR_STATE, the tensor untouched."""
    from liberate_fhe_amd._native import lib, LF_ERR_STATE
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context, twiddles
    ctx = ckks_context(logN=14, num_scales=4, num_special_primes=1, is_secured=False)
    ntt = ntt_context(ctx, devices=["cuda:0"])
    L, N = len(ctx.q), ctx.N
    x = torch.from_numpy(synth.uniform_rows(5, range(L), ctx.q, N, lazy=True)).cuda()
    keep = x.clone()
    psi, ql, qh, kl, kh, q2 = (t[0] for t in (ntt.psi, ntt.ql, ntt.qh, ntt.kl, ntt.kh, ntt._2q))
    st = torch.cuda.current_stream().cuda_stream
    dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
    q = np.array(ctx.q, dtype=np.int64)
    ws = torch.empty((int(lib.lf_ntt_ws_words(1, L, 14)),), dtype=torch.int64, device="cuda")
    one = lambda which: lib.lf_ntt_pass_ws(x.data_ptr(), ws.data_ptr(), 1, L, 14, psi.data_ptr(), dp, q.ctypes.data, 0, 0, which,
                                          ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st)
    try:
        assert one(1) == 0
        lib.lf_tune(4, 0)
        assert one(2) == LF_ERR_STATE
        torch.cuda.synchronize()
        assert torch.equal(x, keep)
        lib.lf_tune(4, 1)
        assert one(2) == 0
        ref = keep.clone()
        assert lib.lf_ntt(ref.data_ptr(), 1, L, 14, psi.data_ptr(), dp, q.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(), qh.data_ptr(),
                          kl.data_ptr(), kh.data_ptr(), 0, st) == 0
        assert torch.equal(x, ref)
    finally:
        lib.lf_tune(4, 1)


@pytest.mark.gpu
def test_c3_silver_cc_mult_decode_within_2_pow_minus_30_of_the_checker():
    """BASELINE configs[2] / SURVEY §8(d) C3: silver cc_mult + relinearize with REAL keys and ciphertexts (HIP samplers)
    on the HIP engine and, on the very same tensors, on the checker engine (reference composition over the C oracle):
    the integer ciphertexts are identical, so the decoded slots differ by less than 2^-30 relative — by exactly 0."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from tests.oracle_backend import OracleBackend
    params = {k: v for k, v in presets.params["silver"].items() if k != "devices"}
    hip = ckks_engine(devices=["cuda:0"], **params)
    chk = ckks_engine(devices=["cpu"], backend=OracleBackend(), **params)

    def to_cpu(x):
        if isinstance(x, torch.Tensor):
            return x.cpu().clone()
        if hasattr(x, "_replace") and hasattr(x, "data"):
            return x._replace(data=to_cpu(x.data))
        if isinstance(x, tuple):
            return tuple(to_cpu(y) for y in x)
        if isinstance(x, list):
            return [to_cpu(y) for y in x]
        return x

    sk = hip.create_secret_key()
    pk, evk = hip.create_public_key(sk), hip.create_evk(sk)
    np.random.seed(11)
    m1, m2 = hip.example(-1, 1), hip.example(-1, 1)
    c1, c2 = hip.encorypt(m1, pk), hip.encorypt(m2, pk)
    prod_hip = hip.cc_mult(c1, c2, evk)
    prod_chk = chk.cc_mult(to_cpu(c1), to_cpu(c2), to_cpu(evk))
    for a, b in zip(prod_hip.data, prod_chk.data):
        assert torch.equal(a[0].cpu(), b[0])
    dec_hip = hip.decrode(prod_hip, sk)
    dec_chk = chk.decrode(prod_chk, to_cpu(sk))
    rel = np.abs(dec_hip - dec_chk).max() / np.abs(dec_chk).max()
    assert rel < 2.0 ** -30, rel
    assert np.abs(dec_hip - m1 * m2).max() < 2e-7        # and both are the product, to CKKS accuracy


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["silver", "sb45"])
def test_hip_native_op_entries_equal_the_step_by_step_engine(name):
    """lf_cc_mult_evk / lf_switch_key (one native call per op over an lf_ks_plan) against the same engine with the entries
    switched off, i.e. the Python orchestration of the individual steps — levels 0, 1 and a deep one, rotate and conjugate, and
    the batched entries (lf_switch_key_batch, lf_cc_mult_evk_batch) against the step-by-step groups."""
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.backend import HipBackend
    params = GOLD[name]["params"]
    outs = []
    for native in (True, False):
        be = HipBackend()
        be.native_ops = native
        eng = ckks_engine(devices=["cuda:0"], backend=be, **params)
        assert (eng._native_level(0) is not None) == native
        evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:3")
        conjk = synth.key_switch_key(eng, 7, origin="conjugation key")
        res = []
        for level in (0, 1, eng.num_levels - 2):
            a, b = synth.ciphertext(eng, 50 + level, level), synth.ciphertext(eng, 60 + level, level)
            prod = eng.cc_mult(a, b, evk)
            res += [digest(prod), digest(eng.rotate_single(a, rotk)), digest(eng.conjugate(a, conjk)),
                    digest(eng.rotate_single(prod, rotk))]
            # the batched entries (lf_switch_key_batch / lf_cc_mult_evk_batch: groups of 4, 2 and a single one) against the
            # Python orchestration of the same groups
            res += [digest(x) for x in eng.rotate_single_batch([a, b, a, a, b, b, a], rotk)]
            res += [digest(x) for x in eng.cc_mult_batch([(a, b), (b, a), (a, a), (b, b), (a, b), (b, a), (a, a)], evk)]
        outs.append(res)
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_key_planes_format_words_and_results():
    """lf_key_planes: the two components of an fp64-class row become one interleaved 32-bit plane + one 16-bit plane of
    the CANONICAL residues, integer-class rows stay raw — checked word for word against numpy on lazy and signed-lazy inputs; and the engine gives the same digests
    whether its key switch reads the key in the planes format (the default for two-pass ring degrees) or raw, for a key
    the engine made, a foreign key, and a key edited in place after its first use."""
    from liberate_fhe_amd.fhe import ckks_engine
    params = dict(logN=13, num_scales=9, num_special_primes=3, is_secured=False)
    eng = ckks_engine(devices=["cuda:0"], **params)
    assert eng._planes_wanted()
    N, c = eng.ctx.N, eng._consts(0, 0, True)
    q = c.q_host
    rng = np.random.default_rng(11)
    src = [np.stack([rng.integers(-int(p) + 1, 2 * int(p), size=N, dtype=np.int64) for p in q]) for _ in range(2)]   # signed-lazy and lazy words
    dst = torch.full((2, len(q), N), -1, dtype=torch.int64, device="cuda:0")
    eng.backend.key_planes(torch.from_numpy(src[0]).cuda(), torch.from_numpy(src[1]).cuda(), dst[0], dst[1], c)
    got = dst.cpu().numpy()
    for r, p in enumerate(q):
        if int(p) >= (1 << 41):
            assert (got[0, r] == src[0][r]).all() and (got[1, r] == src[1][r]).all()
            continue
        cb, ca = src[0][r] % int(p), src[1][r] % int(p)
        lo = got[0, r].view(np.uint32).reshape(N // 2, 4)           # { lo b[j], lo b[j+1], lo a[j], lo a[j+1] }
        hi = got[1, r].view(np.uint16)[:2 * N].reshape(N // 2, 4)   # { hi b[j], hi b[j+1], hi a[j], hi a[j+1] }
        for col, canon in ((0, cb[0::2]), (1, cb[1::2]), (2, ca[0::2]), (3, ca[1::2])):
            assert (lo[:, col] == (canon & 0xffffffff).astype(np.uint32)).all()
            assert (hi[:, col] == (canon >> 32).astype(np.uint16)).all()
        assert (got[1, r].view(np.uint32)[N:] == 0xffffffff).all()          # the rest of the second slot is not written

    def run(e):
        evk, rotk = synth.key_switch_key(e, 5), synth.key_switch_key(e, 6, origin="rotation key:3")     # foreign (unpacked) keys
        sk = e.create_secret_key()
        own = e.create_evk(sk)                                                                          # views of an engine-made pack
        res = []
        for level in (0, 3):
            a, b = synth.ciphertext(e, 50 + level, level), synth.ciphertext(e, 60 + level, level)
            res += [e.cc_mult(a, b, evk), e.rotate_single(a, rotk), e.cc_mult(a, b, own)]
            res += e.rotate_single_batch([a, b, a], rotk) + e.cc_mult_batch([(a, b), (b, a)], own)
        # in-place edit of a used key: the next use must see it (version counters of the pack / the foreign tensors)
        own.data[1].data[0][0][2].add_(1)
        evk.data[0].data[1][0][1].add_(1)
        a, b = synth.ciphertext(e, 70, 1), synth.ciphertext(e, 71, 1)
        res += [e.cc_mult(a, b, own), e.cc_mult(a, b, evk)]
        return [digest(x) for x in res]

    outs = []
    for planes in (True, False):
        e = ckks_engine(devices=["cuda:0"], **params)
        if not planes:
            e._planes_wanted = lambda: False
        # identical randomness for the engine-made key on both sides
        from tests.helpers import SeededCsprng
        e.rng = SeededCsprng(e.ctx.N, [len(di) for di in e.ntt.p.d], max(e.ntt.num_special_primes, 2),
                             devices=list(e.ntt.devices), seed=123)
        outs.append(run(e))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_release_and_invalidate_key_after_a_write_behind_torchs_back():
    """A write into a key through a raw pointer (a native kernel on data_ptr) moves no torch version counter: the engine's
    packed / planes copy of the key is stale until invalidate_key(); release_key drops the copy (rebuilt on the next use) —
    for a foreign key and for a key the engine made."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.fhe import ckks_engine
    params = dict(logN=13, num_scales=9, num_special_primes=2, is_secured=False)
    eng = ckks_engine(devices=["cuda:0"], **params)
    a = synth.ciphertext(eng, 50, 0)
    rows, N = eng.ntt.stops[0][0], eng.ctx.N
    q2 = eng.ntt._2q[0]
    st = torch.cuda.current_stream().cuda_stream
    for make in (lambda: synth.key_switch_key(eng, 6, origin="rotation key:3"),
                 lambda: eng.create_rotation_key(eng.create_secret_key(), 3)):
        key = make()
        before = digest(eng.rotate_single(a, key))
        t = key.data[0].data[0][0]                       # part 0, component b, device 0: [rows, N] lazy words
        ver = t._version
        c = eng._consts(0, 0, True)
        Rs = eng._vec("Rs", 0, 0, True)
        check(lib.lf_mont_enter(t.data_ptr(), Rs.data_ptr(), t.size(0), N, *c.mont(), 0, st), "lf_mont_enter")   # every word x R: other residues
        assert t._version == ver                         # .. and torch saw nothing
        stale = digest(eng.rotate_single(a, key))
        assert stale == before                           # the engine still reads its packed / planes copy: it cannot know
        eng.invalidate_key(key)
        fresh = digest(eng.rotate_single(a, key))
        assert fresh != before                           # rebuilt from the key's tensors as they are now
        eng.release_key(key)
        assert digest(eng.rotate_single(a, key)) == fresh    # dropped and rebuilt once more: same words
        key.data[0].data[1][0][0, 0:8] += 1              # a counted edit: this one the engine sees by itself
        assert digest(eng.rotate_single(a, key)) != fresh


@pytest.mark.gpu
@pytest.mark.parametrize("params", [dict(logN=13, num_scales=9, num_special_primes=2, is_secured=False),
                                    dict(logN=14, num_special_primes=1), dict(logN=13, scale_bits=45, num_scales=6, num_special_primes=2, is_secured=False)])
def test_one_launch_moddown_equals_the_two_launch_form(params):
    """Up to two special primes the mod-down eliminates the special rows inside its single launch (lf_ks_moddown_one over
    constants written once by lf_ks_moddown_consts) instead of a pivots launch in front of it (lf_ks_moddown_ws): same
    words for cc_mult, rotate, conjugate (signed addend) and the batched forms, through the native op entries and through
    the step-by-step orchestration, at two levels."""
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.backend import HipBackend
    outs = []
    for one_max, native in ((2, True), (0, False), (2, False)):
        be = HipBackend()
        be.moddown_one_max_K = one_max
        be.native_ops = native            # (the native entries choose by K themselves: the first round is their one-launch path)
        eng = ckks_engine(devices=["cuda:0"], backend=be, **params)
        evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:3")
        conjk = synth.key_switch_key(eng, 7, origin="conjugation key")
        res = []
        for level in (0, 2):
            a, b = synth.ciphertext(eng, 50 + level, level), synth.ciphertext(eng, 60 + level, level)
            res += [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk), eng.conjugate(a, conjk)]
            res += eng.rotate_single_batch([a, b, a, b, a], rotk) + eng.cc_mult_batch([(a, b), (b, a), (a, a)], evk)
        outs.append([digest(x) for x in res])
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
@pytest.mark.parametrize("params", [dict(logN=13, num_scales=9, num_special_primes=2, is_secured=False), dict(logN=14, num_special_primes=1),
                                    dict(logN=15, num_special_primes=2), dict(logN=13, scale_bits=45, num_scales=6, num_special_primes=4, is_secured=False)])
def test_product_digits_inside_the_last_inverse_pass_equal_the_two_launches(params):
    """cc_mult's x1 * y1: the column thread of the last inverse pass holds the limbs of a digit and runs the Garner step itself
    (lf_intt_mul_digits, where alpha * 2^(logN - 12) <= 32) against inverse transform + ks_digits as two launches
    (lf_tune LF_TUNE_INTT_DIGITS = 0): same words from cc_mult and cc_mult_batch at two levels, digits of 1, 2 and 4 limbs."""
    from liberate_fhe_amd._native import lib
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **params)
    evk = synth.key_switch_key(eng, 5)
    outs = []
    try:
        for fused in (1, 0, 1):
            assert lib.lf_tune(2, fused) in (0, 1)
            res = []
            for level in (0, 2):
                a, b = synth.ciphertext(eng, 50 + level, level), synth.ciphertext(eng, 60 + level, level)
                res += [eng.cc_mult(a, b, evk)] + eng.cc_mult_batch([(a, b), (b, a), (a, a), (b, b), (a, b)], evk)
            outs.append([digest(x) for x in res])
    finally:
        lib.lf_tune(2, 1)
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
@pytest.mark.parametrize("params", [dict(logN=13, num_scales=9, num_special_primes=2, is_secured=False), dict(logN=15, num_special_primes=2),
                                    dict(logN=16, num_special_primes=4), dict(logN=13, scale_bits=45, num_scales=6, num_special_primes=4, is_secured=False)])
def test_key_switch_extension_in_horner_form_equals_the_sum_form(params):
    """The column-form extension evaluates y_0 + m_0 (y_1 + m_1 (..)) over the digit's own primes (engine.ks_horner, the
    default: one modular product per word fewer) or the sum over L_{i-1} y_i: same words from cc_mult, rotate and the batches."""
    from liberate_fhe_amd.fhe import ckks_engine
    outs = []
    for horner in (True, False):
        eng = ckks_engine(devices=["cuda:0"], **params)
        eng.ks_horner = horner
        evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:3")
        res = []
        for level in (0, 2):
            a, b = synth.ciphertext(eng, 50 + level, level), synth.ciphertext(eng, 60 + level, level)
            res += [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
            res += eng.rotate_single_batch([a, b, a, b, a], rotk) + eng.cc_mult_batch([(a, b), (b, a), (a, a)], evk)
        outs.append([digest(x) for x in res])
        tabs = eng._ks_tables(0)
        assert bool(int(tabs[("extend", 0)][0][0, 1]) >> 16) == horner      # the descriptor carries the second table's offset
        del eng
    assert outs[0] == outs[1]


@pytest.mark.gpu
@pytest.mark.parametrize("params", [dict(logN=13, num_scales=9, num_special_primes=2, is_secured=False), dict(logN=14), dict(logN=15, num_special_primes=2),
                                    dict(logN=16, num_special_primes=4), dict(logN=13, scale_bits=45, num_scales=6, num_special_primes=4, is_secured=False),
                                    dict(logN=17, num_scales=11, num_special_primes=3, is_secured=False)])
def test_extended_digits_in_planes_format_equal_raw_words(params):
    """Between the halves of a key switch the fp64-class rows of the scratch hold 6-byte words in two planes (lf_tune
    LF_TUNE_DIGIT_PLANES = 1, the default: ks_ext_cols or the LDS-tiled ks_ext_pass1 (logN 17; LF_TUNE_KS_EXT_COLS_MAX = 0) ->
    fwd_tile16<.., PLN> -> ks_inner2_kernel<.., DPL>) or raw words (0): same
    words from cc_mult, rotate and the batches (launch sets of 4, 2 and 1 ciphertexts) at two levels."""
    from liberate_fhe_amd._native import lib
    from liberate_fhe_amd.fhe import ckks_engine
    eng = ckks_engine(devices=["cuda:0"], **params)
    evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:3")
    outs = []
    try:
        # (more: LF_TUNE_MORE_PLANES — bit 0 the sums between inner product, tiled inverse pass and column pass, bit 1 cc_mult's
        # operand stack between the rescale-NTT, the product pass and the inner product's fold — 6-byte words there as well)
        for planes, cols_max, more in ((1, 5, 3), (0, 5, 3), (1, 0, 3), (0, 0, 0), (1, 5, 0), (1, 5, 1), (1, 5, 2), (1, 0, 1), (1, 5, 3)):
            assert lib.lf_tune(3, planes) in (0, 1)
            lib.lf_tune(1, cols_max)
            assert lib.lf_tune(5, more) in (0, 1, 2, 3)
            res = []
            for level in (0, 2):
                a, b = synth.ciphertext(eng, 50 + level, level), synth.ciphertext(eng, 60 + level, level)
                res += [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
                res += eng.rotate_single_batch([a, b, a, b, a, b, a], rotk) + eng.cc_mult_batch([(a, b), (b, a), (a, a), (b, b), (a, b), (b, a), (a, a)], evk)
            outs.append([digest(x) for x in res])
    finally:
        lib.lf_tune(3, 1)
        lib.lf_tune(1, 5)
        lib.lf_tune(5, 3)
    assert all(o == outs[0] for o in outs)
    assert lib.lf_tune(3, -1) == 1 and lib.lf_tune(3, 2) == 1 and lib.lf_tune(3, -1) == 1      # query; out of range: unchanged


def _reference_shaped_switcher(eng, a, ksk, level):
    """create_switcher as the reference composes it from its step methods (eng.py:746-904) on ONE device: digits per part,
    extension + NTT + key products per part, sums, inverse transform, division by P through the checker's chain."""
    d, N = 0, eng.ctx.N
    rows, ell, K = eng._rows(d, level, True), eng._rows(d, level, False), eng.ntt.num_special_primes
    sums = None
    for part_id in range(len(eng.ntt.p.p[level][d])):
        state = eng.pre_extend(a, d, level, part_id)
        d0, d1 = eng.switcher_later_part(state, ksk, d, d, level, part_id)
        if sums is None:
            sums = [d0, d1]
        else:
            sums = [eng.ntt.mont_add([sums[0]], [d0], level, d, -2)[0], eng.ntt.mont_add([sums[1]], [d1], level, d, -2)[0]]
    s = torch.stack(sums).contiguous()
    eng.ntt.intt_exit_reduce([s[0]], level, d, -2)
    eng.ntt.intt_exit_reduce([s[1]], level, d, -2)
    out = torch.empty((2, ell, N), dtype=torch.int64, device=s.device)
    tabs = eng._ks_tables(level)
    eng.backend.ks_moddown_batch([s[0], s[1]], [out[0], out[1]], [None, None], ell, K, tabs[("pir", d)], eng._vec("Rs", d, level, True),
                                 eng._consts(d, level, True), PiP=None)
    return out


def _check_step_methods(eng):
    ksk = synth.key_switch_key(eng, 21)
    for level in (0, 2):
        a = [synth.ciphertext(eng, 70 + level, level).data[1][0]]
        got = _reference_shaped_switcher(eng, a, ksk, level)
        c0, c1 = eng.create_switcher(a, ksk, level)
        assert torch.equal(got[0], c0[0]) and torch.equal(got[1], c1[0])
    eng.reserve_ksk_buffers()
    assert len(eng.ksk_buffers[0]) == len(eng.ntt.p.p[0][0]) and tuple(eng.ksk_buffers[0][0].shape) == (eng.ntt.num_special_primes, eng.ctx.N)


def test_public_key_switch_step_methods_checker():
    """pre_extend / extend / switcher_later_part / reserve_ksk_buffers (eng.py:218-227, 654-743, 906-937) composed the
    reference's way equal create_switcher (checker backend: host logic)."""
    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    _check_step_methods(ckks_engine(devices=["cpu"], backend=OracleBackend(), **_rot_params()))


@pytest.mark.gpu
@pytest.mark.parametrize("params", [dict(logN=12, num_scales=5, num_special_primes=2, is_secured=False), _rot_params()])
def test_public_key_switch_step_methods_hip(params):
    from liberate_fhe_amd.fhe import ckks_engine
    _check_step_methods(ckks_engine(devices=["cuda:0"], **params))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cols_max", [("gold", 3), ("silver", 0), ("bronze", 0), ("platinum", 4)])
def test_key_switch_extension_column_and_tiled_forms(name, cols_max):
    """The key switch's extension + leading stages as the column kernel (one register step per column) and as the LDS-tiled
    kernel, forced through lf_tune for the size whose default is the other one: the reference digests either way."""
    from liberate_fhe_amd._native import lib
    from liberate_fhe_amd.fhe import ckks_engine
    old = lib.lf_tune(1, -1)
    try:
        lib.lf_tune(1, cols_max)
        rec = GOLD[name]
        check_config(ckks_engine(devices=["cuda:0"], **rec["params"]), rec)
    finally:
        lib.lf_tune(1, old)
