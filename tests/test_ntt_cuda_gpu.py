"""GPU parity of the 15 `ntt_cuda` ops (through the C ABI) against the CPU oracle — bit-exact."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import Limbs, pick_primes, i64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from liberate_fhe_amd.ntt import ntt_cuda
    from oracle import oracle as orc
    return ntt_cuda, orc


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def signed_inputs(lim, seed):
    """Rows mixing canonical, lazy [0,2q), negative (-2q,0) and a few boundary words."""
    rng = np.random.default_rng(seed)
    rows = []
    for x in lim.q:
        v = rng.integers(-2 * x + 1, 2 * x, size=lim.N, dtype=np.int64)
        v[:6] = [0, 1, x - 1, x, 2 * x - 1, -(x - 1)]
        rows.append(v)
    return np.stack(rows)


@pytest.mark.parametrize("logN", [4, 12])
def test_elementwise_ops(mods, logN):
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 3, 2))
    C = lim.rows
    a, b = signed_inputs(lim, 1), signed_inputs(lim, 2)
    d = lambda v: [dev(v)]
    mont = [d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]

    want = np.empty_like(a)
    orc.mont_mult(a, b, want, C, *lim.mont_args())
    got = nc.mont_mult(d(a), d(b), *mont)[0].cpu().numpy()
    assert (got == want).all()

    want = a.copy()
    orc.mont_enter(want, lim.Rs, C, *lim.mont_args())
    t = d(a)
    nc.mont_enter(t, d(lim.Rs), *mont)
    assert (t[0].cpu().numpy() == want).all()

    want = a.copy()
    orc.mont_redc(want, C, *lim.mont_args())
    t = d(a)
    nc.mont_redc(t, *mont)
    assert (t[0].cpu().numpy() == want).all()

    for name in ("reduce_2q", "make_signed", "make_unsigned"):
        want = a.copy()
        getattr(orc, name)(want, C, lim._2q)
        t = d(a)
        getattr(nc, name)(t, d(lim._2q))
        assert (t[0].cpu().numpy() == want).all(), name

    for name in ("mont_add", "mont_sub"):
        want = np.empty_like(a)
        getattr(orc, name)(a, b, want, C, lim._2q)
        got = getattr(nc, name)(d(a), d(b), d(lim._2q))[0].cpu().numpy()
        assert (got == want).all(), name

    vec = a[0].copy()
    want = np.empty_like(a)
    orc.tile_unsigned(vec, want, C, lim._2q)
    src = dev(vec[None, :])
    got = nc.tile_unsigned([src], d(lim._2q))[0]
    assert tuple(got.shape) == (C, lim.N) and (got.cpu().numpy() == want).all()
    assert src.dim() == 1  # squeezed in place like the reference (K.cu:1206)


def test_b_taller_than_a_and_row_extent(mods):
    """mont_mult runs over a.size(0) rows and indexes b / constants by row id (ckks_engine.py:399)."""
    nc, orc = mods
    lim = Limbs(8, pick_primes(8, 3, 1))
    a, b = lim.uniform(3, lazy=True)[:3].copy(), lim.uniform(4, lazy=True)
    want = np.empty_like(a)
    orc.mont_mult(a, b, want, 3, *lim.mont_args())
    got = nc.mont_mult([dev(a)], [dev(b)], [dev(lim.ql)], [dev(lim.qh)], [dev(lim.kl)], [dev(lim.kh)])[0]
    assert (got.cpu().numpy() == want).all()


def test_rescale_known_answer(mods):
    """REDC of a negative difference must return the NEGATIVE representative."""
    nc, orc = mods
    q_l, q_i = 1099510054913, 1099515691009
    d_, s_ = 49326798554, 1033533601499
    k = (2**62 * pow(2**62, -1, q_i) - 1) // q_i
    scale = pow(q_l, -1, q_i) * 2**62 % q_i
    a = dev(np.full((1, 16), d_ - s_, dtype=np.int64))
    lb = (1 << 31) - 1
    t = lambda v: [dev(i64([v]))]
    nc.mont_enter([a], t(scale), t(q_i & lb), t(q_i >> 31), t(k & lb), t(k >> 31))
    assert (a.cpu().numpy() == -20459).all()


@pytest.mark.parametrize("logN", [1, 2, 3, 4, 5, 6, 7, 9, 11, 12, 13, 14, 15, 16, 17])
def test_ntt_family_bit_exact(mods, logN):
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    C = lim.rows
    psi, ipsi = lim.mont_tables()
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    x = lim.uniform(10 + logN, lazy=True)

    # ntt
    want = x.copy()
    orc.ntt(want, psi, C, logN, lim._2q, *lim.mont_args())
    t = d(x)
    nc.ntt(t, [None], [None], d(psi), *consts)
    fwd = t[0].cpu().numpy()
    assert (fwd == want).all(), "ntt"

    # enter_ntt
    xc = lim.uniform(20 + logN)
    want_e = xc.copy()
    orc.mont_enter(want_e, lim.Rs, C, *lim.mont_args())
    orc.ntt(want_e, psi, C, logN, lim._2q, *lim.mont_args())
    t = d(xc)
    nc.enter_ntt(t, d(lim.Rs), [None], [None], d(psi), *consts)
    assert (t[0].cpu().numpy() == want_e).all(), "enter_ntt"

    # the four inverse chains
    for tail, name in enumerate(("intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed")):
        want_i = want_e.copy()
        orc.intt(want_i, ipsi, lim.Ninv, C, logN, lim._2q, *lim.mont_args())
        if tail >= 1:
            orc.mont_redc(want_i, C, *lim.mont_args())
        if tail >= 2:
            orc.reduce_2q(want_i, C, lim._2q)
        if tail >= 3:
            orc.make_signed(want_i, C, lim._2q)
        t = d(want_e)
        getattr(nc, name)(t, [None], [None], d(ipsi), d(lim.Ninv), *consts)
        assert (t[0].cpu().numpy() == want_i).all(), name
        if tail == 2:
            assert (want_i == xc).all(), "round trip must be the identity"


@pytest.mark.parametrize("logN", [12, 13, 16])
def test_ntt_tiles_with_out_of_range_words(mods, logN):
    """A 4096-word tile holding a word outside [0, 2q) leaves the register-fed fp64 form before anything is
    stored and is redone by the signed integer routine; its neighbours stay on the fast form.  Both must equal
    the oracle (which follows the reference's signed arithmetic literally)."""
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    C = lim.rows
    psi, ipsi = lim.mont_tables()
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    x = lim.uniform(77 + logN, lazy=True)
    for r, q in enumerate(lim.q):
        x[r, 5] = -(q - 3)                 # negative lazy word in the first tile
        x[r, lim.N - 4096 + 4095] = 2 * q + 1 if r != 1 else x[r, lim.N - 1]   # one row keeps a clean last tile
        if lim.N > 8192:
            x[r, 4096 + 513] = 2 * q       # boundary word, second tile
    want = x.copy()
    orc.ntt(want, psi, C, logN, lim._2q, *lim.mont_args())
    t = d(x)
    nc.ntt(t, [None], [None], d(psi), *consts)
    assert (t[0].cpu().numpy() == want).all(), "ntt"

    y = lim.uniform(78 + logN, lazy=True)
    for r, q in enumerate(lim.q):
        y[r, 9] = -1
        y[r, lim.N - 7] = 2 * q + 5
    for tail, name in enumerate(("intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed")):
        want_i = y.copy()
        orc.intt(want_i, ipsi, lim.Ninv, C, logN, lim._2q, *lim.mont_args())
        if tail >= 1:
            orc.mont_redc(want_i, C, *lim.mont_args())
        if tail >= 2:
            orc.reduce_2q(want_i, C, lim._2q)
        if tail >= 3:
            orc.make_signed(want_i, C, lim._2q)
        t = d(y)
        getattr(nc, name)(t, [None], [None], d(ipsi), d(lim.Ninv), *consts)
        assert (t[0].cpu().numpy() == want_i).all(), name


@pytest.mark.parametrize("logN", [13, 14, 15, 16, 17])
def test_transforms_through_a_workspace_equal_in_place_and_oracle(mods, logN):
    """lf_ntt_ws / lf_intt_ws (what ntt_cuda.ntt / enter_ntt / intt* call at logN 13 .. 17: the two passes exchange the fp64-class
    limbs as 6-byte planes through a workspace) against lf_ntt / lf_intt (USE_WORKSPACE = False) and the oracle: lazy operands; operands with arbitrary
    words below 2^61 in magnitude sprinkled over columns and tiles (a column wave that meets one ships its words' top 16 bits in a third
    plane and raises its flag; the tiles behind leave the fast form); a workspace full of ones before every call."""
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 3, 2))
    C = lim.rows
    psi, ipsi = lim.mont_tables()
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    rng = np.random.default_rng(900 + logN)
    clean = lim.uniform(31 + logN, lazy=True)
    dirty = clean.copy()
    for r, q in enumerate(lim.q):
        cols = rng.integers(0, lim.N, size=24)
        dirty[r, cols] = rng.integers(-(2 ** 61), 2 ** 61, size=24, dtype=np.int64)
        dirty[r, 64 * r + 1] = 2 * q                 # boundary word
        dirty[r, lim.N - 1 - 64 * r] = -1
    one_row_clean = dirty.copy()
    one_row_clean[1] = clean[1]                      # its flags must read 0 beside flagged neighbours
    wild = dirty.copy()                              # any int64 at all: beyond the oracle's defined range (signed overflow in C),
    for r in range(C):                               # the two forms are compared with each other
        wild[r, rng.integers(0, lim.N, size=24)] = rng.integers(-(2 ** 63), 2 ** 63 - 1, size=24, dtype=np.int64)
    assert nc.USE_WORKSPACE
    try:
        both = {}
        for use in (True, False):
            nc.USE_WORKSPACE = use
            t = d(wild)
            nc.enter_ntt(t, d(lim.Rs), [None], [None], d(psi), *consts)
            both[use] = t[0].cpu().numpy()
        assert (both[True] == both[False]).all(), "any int64 words: workspace against in place"
        for use in (True, False):
            nc.USE_WORKSPACE = use
            t = d(wild)
            nc.intt_exit(t, [None], [None], d(ipsi), d(lim.Ninv), *consts)
            both[use] = t[0].cpu().numpy()
        assert (both[True] == both[False]).all(), "any int64 words, inverse: workspace against in place"
        for name, x in (("lazy", clean), ("arbitrary words", dirty), ("one clean limb", one_row_clean)):
            for entry in ("ntt", "enter_ntt"):
                want = x.copy()
                if entry == "enter_ntt":
                    orc.mont_enter(want, lim.Rs, C, *lim.mont_args())
                orc.ntt(want, psi, C, logN, lim._2q, *lim.mont_args())
                got = {}
                for use in (True, False):
                    nc.USE_WORKSPACE = use
                    for ws in nc._WS.values():
                        ws.fill_(-1)
                    t = d(x)
                    if entry == "ntt":
                        nc.ntt(t, [None], [None], d(psi), *consts)
                    else:
                        nc.enter_ntt(t, d(lim.Rs), [None], [None], d(psi), *consts)
                    got[use] = t[0].cpu().numpy()
                assert (got[True] == want).all(), f"{entry}, {name}: through the workspace"
                assert (got[False] == want).all(), f"{entry}, {name}: in place"
            for tail, chain in enumerate(("intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed")):
                want = x.copy()
                orc.intt(want, ipsi, lim.Ninv, C, logN, lim._2q, *lim.mont_args())
                if tail >= 1:
                    orc.mont_redc(want, C, *lim.mont_args())
                if tail >= 2:
                    orc.reduce_2q(want, C, lim._2q)
                if tail >= 3:
                    orc.make_signed(want, C, lim._2q)
                for use in (True, False):
                    nc.USE_WORKSPACE = use
                    for ws in nc._WS.values():
                        ws.fill_(-1)
                    t = d(x)
                    getattr(nc, chain)(t, [None], [None], d(ipsi), d(lim.Ninv), *consts)
                    assert (t[0].cpu().numpy() == want).all(), f"{chain}, {name}: {'through the workspace' if use else 'in place'}"
    finally:
        nc.USE_WORKSPACE = True
    assert any(ws.numel() >= C * lim.N for ws in nc._WS.values())


@pytest.mark.parametrize("logN", [13, 14, 15, 16])
def test_workspace_split_with_and_without_the_extra_column_stage(mods, logN):
    """lf_ntt_ws splits a transform as (logN - 12 + 1) column stages + 11 tile stages (LF_TUNE_WS_EXTRA_STAGE = 1, the default: the
    tiles skip their first stage, flags are indexed by 2048-wide columns) or as lf_ntt does (0): the oracle's words either way,
    with out-of-range words in both halves of a tile."""
    from liberate_fhe_amd._native import lib
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    C = lim.rows
    psi, _ = lim.mont_tables()
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    x = lim.uniform(61 + logN, lazy=True)
    for r, q in enumerate(lim.q):
        x[r, 100 + r] = 2 * q + 1                    # first half of tile 0
        x[r, 2048 + 300 + r] = -5                    # second half of tile 0: the same column wave's partner word
        x[r, lim.N - 1] = 3 * q
    want = x.copy()
    orc.mont_enter(want, lim.Rs, C, *lim.mont_args())
    orc.ntt(want, psi, C, logN, lim._2q, *lim.mont_args())
    assert lib.lf_tune(4, -1) == 1
    try:
        for extra in (1, 0, 1):
            assert lib.lf_tune(4, extra) in (0, 1)
            for ws in nc._WS.values():
                ws.fill_(-1)
            t = d(x)
            nc.enter_ntt(t, d(lim.Rs), [None], [None], d(psi), *consts)
            assert (t[0].cpu().numpy() == want).all(), f"extra column stage {extra}"
    finally:
        lib.lf_tune(4, 1)


@pytest.mark.parametrize("small,large", [(3, 0), (0, 2), (1, 0)])
def test_workspace_transforms_with_one_arithmetic_class_only(mods, small, large):
    """The workspace kernels take both class lists in one launch; either may be empty (only 40-bit primes / only 60-bit primes)."""
    nc, orc = mods
    for logN in (13, 16):
        lim = Limbs(logN, pick_primes(logN, small, large))
        C = lim.rows
        psi, ipsi = lim.mont_tables()
        d = lambda v: [dev(v)]
        consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
        x = lim.uniform(41 + logN)
        x[0, 7] = 2 * lim.q[0] + 3                       # one word outside [0, 2q)
        want = x.copy()
        orc.mont_enter(want, lim.Rs, C, *lim.mont_args())
        orc.ntt(want, psi, C, logN, lim._2q, *lim.mont_args())
        t = d(x)
        nc.enter_ntt(t, d(lim.Rs), [None], [None], d(psi), *consts)
        assert (t[0].cpu().numpy() == want).all(), (logN, "enter_ntt")
        back = want.copy()
        orc.intt(back, ipsi, lim.Ninv, C, logN, lim._2q, *lim.mont_args())
        orc.mont_redc(back, C, *lim.mont_args())
        orc.reduce_2q(back, C, lim._2q)
        nc.intt_exit_reduce(t, [None], [None], d(ipsi), d(lim.Ninv), *consts)
        assert (t[0].cpu().numpy() == back).all(), (logN, "intt_exit_reduce")


def test_workspace_is_per_stream(mods):
    """ntt_cuda keeps one workspace per (device, stream): transforms enqueued on two streams at once use two buffers and both
    equal the oracle (one shared buffer would be written by both column passes before either tiled pass read it)."""
    nc, orc = mods
    logN = 14
    lim = Limbs(logN, pick_primes(logN, 3, 1))
    psi, _ = lim.mont_tables()
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    xs = [lim.uniform(70 + i, lazy=True) for i in range(2)]
    wants = []
    for x in xs:
        w = x.copy()
        orc.ntt(w, psi, lim.rows, logN, lim._2q, *lim.mont_args())
        wants.append(w)
    table = d(psi)
    nc.ntt(d(xs[0]), [None], [None], table, *consts)     # builds the table's auxiliary twin once, on the default stream
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for _ in range(8):                                   # several rounds of two transforms in flight
        ts = [d(x) for x in xs]
        torch.cuda.synchronize()
        for st, t in zip(streams, ts):
            with torch.cuda.stream(st):
                nc.ntt(t, [None], [None], table, *consts)
        torch.cuda.synchronize()
        outs.append(ts)
    for ts in outs:
        for t, w in zip(ts, wants):
            assert (t[0].cpu().numpy() == w).all()
    keys = {(0, st.cuda_stream) for st in streams}
    assert keys <= set(nc._WS), "one workspace per stream"
    assert nc._WS[(0, streams[0].cuda_stream)].data_ptr() != nc._WS[(0, streams[1].cuda_stream)].data_ptr()


def test_workspace_entry_argument_checks(mods):
    """lf_ntt_ws: relaxed transforms are refused before anything is launched; without a workspace, and at sizes with one
    launch (logN <= 12), it IS lf_ntt."""
    from liberate_fhe_amd._native import lib
    from liberate_fhe_amd.ntt import twiddles
    nc, orc = mods
    for logN in (12, 13, 17):
        lim = Limbs(logN, pick_primes(logN, 2, 1))
        psi, _ = lim.mont_tables()
        x = lim.uniform(3, lazy=True)
        want = x.copy()
        orc.ntt(want, psi, lim.rows, logN, lim._2q, *lim.mont_args())
        c = [dev(v) for v in (psi, lim.ql, lim.qh, lim.kl, lim.kh)]
        st = torch.cuda.current_stream().cuda_stream
        dp = twiddles.dp_pointer(c[0], c[1], c[2], c[3], c[4], 0, st)
        qh = np.asarray(lim.q, dtype=np.int64)
        words = lib.lf_ntt_ws_words(1, lim.rows, logN)
        assert words == lim.rows * lim.N + 8 * lim.rows
        ws = torch.full((words,), -1, dtype=torch.int64, device="cuda")
        for wsp in (ws.data_ptr(), 0):
            t = dev(x)
            assert lib.lf_ntt_ws(t.data_ptr(), wsp, 1, lim.rows, logN, c[0].data_ptr(), dp, qh.ctypes.data, 0, 0,
                                 c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4].data_ptr(), 0, st) == 0
            assert (t.cpu().numpy() == want).all(), (logN, wsp != 0)
        t = dev(x)
        assert lib.lf_ntt_ws(t.data_ptr(), ws.data_ptr(), 1, lim.rows, logN, c[0].data_ptr(), dp, qh.ctypes.data, 0, 1,
                             c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4].data_ptr(), 0, st) != 0
        torch.cuda.synchronize()
        assert (t.cpu().numpy() == x).all()          # refused: nothing ran
    assert lib.lf_ntt_ws_words(-1, 1, 13) == -1 and lib.lf_ntt_ws_words(1, 1, 99) == -1


def test_ntt_extent_is_constant_rows(mods):
    """NTT-family ops transform ql.size(0) rows and leave further rows of `a` alone (K.cu:298)."""
    nc, orc = mods
    logN = 12
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    psi, _ = lim.mont_tables()
    x = lim.uniform(5, lazy=True)
    t = dev(x)
    cut = lambda v: [dev(v[:2])]
    nc.ntt([t], [None], [None], cut(psi), cut(lim._2q), cut(lim.ql), cut(lim.qh), cut(lim.kl), cut(lim.kh))
    got = t.cpu().numpy()
    want = x.copy()
    orc.ntt(want[:2], psi[:2].copy(), 2, logN, lim._2q[:2], lim.ql[:2], lim.qh[:2], lim.kl[:2], lim.kh[:2])
    assert (got == want).all()


def test_reference_shaped_tables_are_accepted(mods):
    """Drop-in: the reference passes [rows, logN, N/2] per-stage tables; results must be identical."""
    nc, orc = mods
    from liberate_fhe_amd.fhe.context.ckks_context import stage_butterfly_indices
    logN = 10
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    psi, ipsi = lim.mont_tables()
    ev, od, tw = stage_butterfly_indices(logN, inverse=False)
    iev, iod, itw = stage_butterfly_indices(logN, inverse=True)
    psi3, ipsi3 = np.ascontiguousarray(psi[:, tw]), np.ascontiguousarray(ipsi[:, itw])
    d = lambda v: [dev(v)]
    consts = [d(lim._2q), d(lim.ql), d(lim.qh), d(lim.kl), d(lim.kh)]
    x = lim.uniform(9, lazy=True)
    a, b = d(x), d(x)
    nc.ntt(a, d(ev), d(od), d(psi3), *consts)
    nc.ntt(b, [None], [None], d(psi), *consts)
    assert torch.equal(a[0], b[0])
    want = x.copy()
    orc.ntt_tab(want, ev, od, psi3, lim.rows, lim._2q, *lim.mont_args())
    assert (a[0].cpu().numpy() == want).all()
    nc.intt(a, d(iev), d(iod), d(ipsi3), d(lim.Ninv), *consts)
    nc.intt(b, [None], [None], d(ipsi), d(lim.Ninv), *consts)
    assert torch.equal(a[0], b[0])
    orc.intt_tab(want, iev, iod, ipsi3, lim.Ninv, lim.rows, lim._2q, *lim.mont_args())
    assert (a[0].cpu().numpy() == want).all()


@pytest.mark.parametrize("logN", [12, 13, 15, 16])
@pytest.mark.parametrize("flags", [0, 1, 3])
def test_rescale_ntt_equals_rescale_then_ntt(logN, flags):
    """lf_rescale_ntt (rescale inside the first NTT pass for logN 13..16) against lf_rescale_batch + lf_ntt, both of
    which are pinned to the oracle above: same words, exact / relaxed / relaxed+plain."""
    import ctypes
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.ntt import twiddles
    lim = Limbs(logN, pick_primes(logN, 3, 2))
    rows, N, count = lim.rows, lim.N, 4
    rng = np.random.default_rng(logN * 10 + flags)
    q_drop = int(lim.q[0]) | 1
    srcs = [dev(np.stack([rng.integers(0, int(x), size=N, dtype=np.int64) for x in lim.q])) for _ in range(count)]
    row0 = [dev(rng.integers(0, q_drop, size=N, dtype=np.int64)) for _ in range(count)]
    scales = dev(np.array([rng.integers(1, int(x)) for x in lim.q], dtype=np.int64))
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    psi, Rs, q2 = dev(lim.mont_tables()[0]), dev(lim.Rs), dev(lim._2q)
    st = torch.cuda.current_stream().cuda_stream
    dp = twiddles.dp_pointer(psi, *c, 0, st)
    q_host = np.array(lim.q, dtype=np.int64)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    want = torch.empty((count, rows, N), dtype=torch.int64, device="cuda")
    check(lib.lf_rescale_batch(arr(srcs), arr(row0), arr([want[i] for i in range(count)]), count, rows, N, scales.data_ptr(),
                               q_drop // 2, *[t.data_ptr() for t in c], 0, st), "rescale")
    check(lib.lf_ntt(want.data_ptr(), count, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, Rs.data_ptr(), flags,
                     q2.data_ptr(), *[t.data_ptr() for t in c], 0, st), "ntt")
    got = torch.empty_like(want)
    check(lib.lf_rescale_ntt(arr(srcs), arr(row0), count, got.data_ptr(), rows, logN, scales.data_ptr(), q_drop // 2,
                             psi.data_ptr(), dp, q_host.ctypes.data, Rs.data_ptr(), flags, q2.data_ptr(),
                             *[t.data_ptr() for t in c], 0, st), "rescale_ntt")
    assert torch.equal(got, want)


def test_batch_entry_points_edge_cases():
    """count = 0 is a no-op, count > 8 and bad Galois exponents are argument errors, single-set batches equal the
    scalar entry points."""
    import ctypes
    from liberate_fhe_amd._native import lib
    logN = 12
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    rows, N = lim.rows, lim.N
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    cp = [t.data_ptr() for t in c]
    st = torch.cuda.current_stream().cuda_stream
    none = (ctypes.c_void_p * 1)()
    scales = dev(np.ones(rows, dtype=np.int64))
    assert lib.lf_rescale_batch(none, none, none, 0, rows, N, scales.data_ptr(), 1, *cp, 0, st) == 0
    nine = (ctypes.c_void_p * 9)()
    assert lib.lf_rescale_batch(nine, nine, nine, 9, rows, N, scales.data_ptr(), 1, *cp, 0, st) == 10001
    assert lib.lf_galois_batch(nine, nine, 9, rows, logN, 5, 0, 0, st) == 10001
    a = dev(lim.uniform(1))
    d1, d2 = torch.empty_like(a), torch.empty_like(a)
    q2 = dev(lim._2q)
    arr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
    assert lib.lf_galois_batch(arr(a), arr(a), 1, rows, logN, 5, 0, 0, st) == 10001            # in place
    assert lib.lf_galois_batch(arr(a), arr(d1), 1, rows, logN, 4, 0, 0, st) == 10001           # even exponent
    assert lib.lf_galois(a.data_ptr(), d1.data_ptr(), rows, logN, 5, q2.data_ptr(), 0, st) == 0
    assert lib.lf_galois_batch(arr(a), arr(d2), 1, rows, logN, 5, q2.data_ptr(), 0, st) == 0
    assert torch.equal(d1, d2)


@pytest.mark.parametrize("canonical", [True, False])
def test_galois_gather_equals_scatter(canonical):
    """lf_ks_digits_galois / the gather addend of lf_ks_moddown_batch read a(X^p) exactly as lf_galois writes it:
    checked on the digit kernel with a single one-limb digit (its Garner state is the input word itself)."""
    from liberate_fhe_amd._native import lib, check
    logN = 12
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    rows, N = lim.rows, lim.N
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    cp = [t.data_ptr() for t in c]
    st = torch.cuda.current_stream().cuda_stream
    a = dev(lim.uniform(7))
    q2 = dev(lim._2q)
    p = pow(3, 5, 2 * N)
    want = torch.empty_like(a)
    check(lib.lf_galois(a.data_ptr(), want.data_ptr(), rows, logN, p, q2.data_ptr() if canonical else 0, 0, st), "galois")
    desc = dev(np.array([[r, 1, 0, 0] for r in range(rows)], dtype=np.int64))     # one single-limb digit per row
    tab = dev(np.zeros(1, dtype=np.int64))
    got = torch.empty_like(a)
    check(lib.lf_ks_digits_galois(a.data_ptr(), got.data_ptr(), rows, desc.data_ptr(), tab.data_ptr(), N, pow(p, -1, 2 * N),
                                  q2.data_ptr() if canonical else 0, *cp, 0, st), "digits_galois")
    assert torch.equal(got, want)


def test_ntt_pass_entry_is_the_two_halves_of_lf_ntt():
    """lf_ntt_pass (the measurement entry bench.py times): which = 1 then which = 2 on one buffer == lf_ntt;
    single-pass sizes and other selectors are argument errors."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.ntt import twiddles
    logN = 14
    lim = Limbs(logN, pick_primes(logN, 3, 2))
    rows, batch = lim.rows, 3
    psi, q2 = dev(lim.mont_tables()[0]), dev(lim._2q)
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    cp = [t.data_ptr() for t in c]
    st = torch.cuda.current_stream().cuda_stream
    dp = twiddles.dp_pointer(psi, *c, 0, st)
    q_host = np.array(lim.q, dtype=np.int64)
    x = dev(np.stack([lim.uniform(40 + b, lazy=True) for b in range(batch)]))
    want, got = x.clone(), x.clone()
    check(lib.lf_ntt(want.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), *cp, 0, st), "ntt")
    for which in (1, 2):
        check(lib.lf_ntt_pass(got.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, which, *cp, 0, st), "pass")
    assert torch.equal(got, want)
    assert lib.lf_ntt_pass(got.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, 3, *cp, 0, st) == 10001
    assert lib.lf_ntt_pass(got.data_ptr(), batch, rows, 12, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, 1, *cp, 0, st) == 10001


@pytest.mark.parametrize("logN", [13, 14, 15, 16])
@pytest.mark.parametrize("flags", [0, 1])
def test_large_batch_transform_equals_its_two_passes_and_the_oracle(mods, logN, flags):
    """Large batches (16 384+ tiles, an odd polynomial count): whatever launch shapes lf_ntt picks for them (8 tiles per
    block from 32 768 tiles up) must give the words of the two plain passes launched one after the other (lf_ntt_pass,
    which the previous test ties to lf_ntt at small batch), and the oracle's on the first, a middle and the last
    polynomial — including tiles that leave the fast form (signed words) inside the big launch."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.ntt import twiddles
    nc, orc = mods
    lim = Limbs(logN, pick_primes(logN, 3, 2))
    rows = lim.rows
    min_polys = -(-4096 // (rows << (logN - 12)))
    batch = 4 * min_polys + 3
    psi_np = lim.mont_tables()[0]
    psi, q2 = dev(psi_np), dev(lim._2q)
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    cp = [t.data_ptr() for t in c]
    st = torch.cuda.current_stream().cuda_stream
    dp = twiddles.dp_pointer(psi, *c, 0, st)
    q_host = np.array(lim.q, dtype=np.int64)
    g = torch.Generator(device="cuda").manual_seed(logN * 7 + flags)
    x = torch.empty((batch, rows, lim.N), dtype=torch.int64, device="cuda")
    for r, q in enumerate(lim.q):
        x[:, r] = torch.randint(0, 2 * q, (batch, lim.N), generator=g, device="cuda", dtype=torch.int64)
    picks = [0, batch // 2, batch - 1]
    if not flags:
        for p in picks:                              # signed-lazy words: those tiles take the integer routine
            for r, q in enumerate(lim.q):
                x[p, r, 5] = -(q - 3)
                x[p, r, lim.N - 9] = 2 * q + 1
    want, got = x.clone(), x.clone()
    for which in (1, 2):
        check(lib.lf_ntt_pass(want.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, flags, which, *cp, 0, st), "pass")
    check(lib.lf_ntt(got.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, flags, q2.data_ptr(), *cp, 0, st), "ntt")
    assert torch.equal(got, want)
    for p in picks:
        ref = x[p].cpu().numpy().copy()
        orc.ntt(ref, psi_np, rows, logN, lim._2q, *lim.mont_args())
        out = got[p].cpu().numpy()
        if flags:                                    # relaxed: canonical residues of the same values
            ref = ref % q_host[:, None]
            out = out % q_host[:, None]
        assert (out == ref).all(), p


def test_auxiliary_table_layout_and_shoup_quotients():
    """lf_twiddle_dp: rows of 2N words — plain twiddles as doubles (slot 0: 1/q) for primes below 2^41, pairs
    (floor(w 2^64 / q), w) for the integer class — against big-integer arithmetic on the plain table."""
    from liberate_fhe_amd._native import lib, check
    logN = 13
    lim = Limbs(logN, pick_primes(logN, 2, 2))
    N = lim.N
    psi = dev(lim.mont_tables()[0])
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    out = torch.zeros((lim.rows, 2 * N), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib.lf_twiddle_dp(psi.data_ptr(), out.data_ptr(), lim.rows, N, *[t.data_ptr() for t in c], 0, st), "twiddle_dp")
    host = out.cpu()
    for r, q in enumerate(lim.q):
        q = int(q)
        plain = [int(v) for v in lim.psi_plain[r]]
        if q < (1 << 41):
            got = host[r, :N].numpy()
            assert got[0] == 1.0 / q and (got[1:] == np.array(plain[1:], dtype=np.float64)).all()
        else:
            pairs = host[r].view(torch.int64).numpy().view(np.uint64).reshape(N, 2)
            for j in (1, 2, 3, N // 2, N - 1):
                assert int(pairs[j, 1]) == plain[j] and int(pairs[j, 0]) == (plain[j] << 64) // q, (r, j)


@pytest.mark.parametrize("logN", [13, 14, 15, 16])
def test_relaxed_transforms_integer_class_shoup_round_trip(mods, logN):
    """The relaxed transforms run the integer class on Shoup products (lazy 64-bit words, canonical on the way out of
    every pass): forward == the oracle's transform modulo q, forward then inverse (tail 2) == the input; 60-bit rows
    alone and next to fp64-class rows; inputs at the edges of the accepted range."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.ntt import twiddles
    nc, orc = mods
    for n40, n60 in ((0, 3), (2, 2)):
        lim = Limbs(logN, pick_primes(logN, n40, n60))
        rows, batch = lim.rows, 3
        psi_np, ipsi_np = lim.mont_tables()
        psi, ipsi, q2, Ninv = dev(psi_np), dev(ipsi_np), dev(lim._2q), dev(lim.Ninv)
        c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
        cp = [t.data_ptr() for t in c]
        st = torch.cuda.current_stream().cuda_stream
        dp, idp = twiddles.dp_pointer(psi, *c, 0, st), twiddles.dp_pointer(ipsi, *c, 0, st)
        q_host = np.array(lim.q, dtype=np.int64)
        x_np = np.stack([lim.uniform(900 + logN + b) for b in range(batch)])
        for r, q in enumerate(lim.q):
            x_np[0, r, :4] = [0, 1, q - 1, q - 2]
            x_np[1, r, :] = q - 1                      # every lazy sum at its upper bound
            x_np[1, r, 1::3] = 0                       # ... and every difference too
        x = dev(x_np)
        y = x.clone()
        check(lib.lf_ntt(y.data_ptr(), batch, rows, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, 1, q2.data_ptr(), *cp, 0, st), "ntt")
        got = y.cpu().numpy()
        assert (got >= 0).all() and (got < q_host[None, :, None]).all()          # canonical residues
        for b in range(batch):
            ref = x_np[b].copy()
            orc.ntt(ref, psi_np, rows, logN, lim._2q, *lim.mont_args())
            assert (got[b] == ref % q_host[:, None]).all(), (n40, n60, b)
        # lazy words in [0, 2q) are what the inverse accepts (include/ckks_hip.h)
        z = y + torch.from_numpy(q_host)[None, :, None].cuda() * (torch.arange(lim.N, device="cuda") % 2)[None, None, :]
        # the reference chain needs Montgomery-form input for intt_exit_reduce to return x: enter first
        for t in (y, z):
            w = t.clone()
            check(lib.lf_intt(w.data_ptr(), batch, rows, logN, ipsi.data_ptr(), idp, q_host.ctypes.data, Ninv.data_ptr(), 2, 1,
                              q2.data_ptr(), *cp, 0, st), "intt")
            for b in range(batch):
                ref = got[b].copy()
                orc.intt(ref, ipsi_np, lim.Ninv, rows, logN, lim._2q, *lim.mont_args())
                orc.mont_redc(ref, rows, *lim.mont_args())
                orc.reduce_2q(ref, rows, lim._2q)
                assert (w[b].cpu().numpy() == ref).all(), (n40, n60, b)


def test_relaxed_transforms_require_the_auxiliary_table():
    """LF_NTT_RELAXED without psi_dp / ipsi_dp is an argument error (the relaxed arithmetic of both classes lives in that
    table), reported before anything is launched: the buffer is untouched."""
    from liberate_fhe_amd._native import lib
    logN = 13
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    psi, ipsi = (dev(t) for t in lim.mont_tables())
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    cp = [t.data_ptr() for t in c]
    q2, Ninv = dev(lim._2q), dev(lim.Ninv)
    st = torch.cuda.current_stream().cuda_stream
    q_host = np.array(lim.q, dtype=np.int64)
    x = dev(lim.uniform(3))
    keep = x.clone()
    assert lib.lf_ntt(x.data_ptr(), 1, lim.rows, logN, psi.data_ptr(), 0, q_host.ctypes.data, 0, 1, q2.data_ptr(), *cp, 0, st) == 10001
    assert lib.lf_intt(x.data_ptr(), 1, lim.rows, logN, ipsi.data_ptr(), 0, q_host.ctypes.data, Ninv.data_ptr(), 2, 1, q2.data_ptr(), *cp, 0, st) == 10001
    torch.cuda.synchronize()
    assert torch.equal(x, keep)


def test_integration_md_binding_snippet_runs():
    """The binding INTEGRATION.md (path A) shows a reference maintainer — taken from the file, executed as written (only the
    library path is resolved) — gives what the tested shim gives, on reference-shaped [rows, logN, N/2] twiddle tables."""
    import re
    from liberate_fhe_amd import _native
    from liberate_fhe_amd.ntt import ntt_cuda
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# src/liberate/ntt/ntt_cuda\.py.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libckks_hip.so")', f'ctypes.CDLL({_native.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    logN = 12
    lim = Limbs(logN, pick_primes(logN, 2, 1))
    N = lim.N
    from liberate_fhe_amd.fhe.context.ckks_context import stage_butterfly_indices
    _, _, tw = stage_butterfly_indices(logN, inverse=False)
    stage = np.ascontiguousarray(lim.mont_tables()[0][:, tw])      # the reference's [rows, logN, N/2] per-stage layout
    assert stage.shape == (lim.rows, logN, N // 2)
    c = [dev(v) for v in (lim.ql, lim.qh, lim.kl, lim.kh)]
    a, b = dev(lim.uniform(5, lazy=True)), dev(lim.uniform(6, lazy=True))
    got = ns["mont_mult"]([a], [b], [c[0]], [c[1]], [c[2]], [c[3]])[0]
    want = ntt_cuda.mont_mult([a], [b], [c[0]], [c[1]], [c[2]], [c[3]])[0]
    assert torch.equal(got, want)
    x1, x2 = a.clone(), a.clone()
    psi_s, q2 = dev(stage), dev(lim._2q)
    ns["ntt"]([x1], None, None, [psi_s], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
    ntt_cuda.ntt([x2], None, None, [psi_s], [q2], [c[0]], [c[1]], [c[2]], [c[3]])
    assert torch.equal(x1, x2)


@pytest.mark.gpu
def test_clock_probe_reports_a_plausible_shader_clock():
    """lf_clock_probe (measurement entry of the C ABI): cycles per 100 MHz tick on an idle device -> 0.5 .. 3.5 GHz."""
    import torch
    from liberate_fhe_amd._native import lib, check
    out = torch.zeros(2 * 4, dtype=torch.int64, device="cuda:0")
    check(lib.lf_clock_probe(out.data_ptr(), 4, 20_000, 0, torch.cuda.current_stream().cuda_stream), "lf_clock_probe")
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(4, 2)
    assert (o[:, 1] >= 20_000).all()
    mhz = o[:, 0] / o[:, 1] * 100.0
    assert ((mhz > 500) & (mhz < 3500)).all(), mhz
