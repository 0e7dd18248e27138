"""Generate tests/golden/engine_digests.json by running the REFERENCE engine (imported from
/root/reference, CUDA modules replaced by the C oracle — see refdriver.py) on synthetic inputs.

    python tests/golden/make_golden.py            # build container only

Inputs are regenerated from seeds by liberate_fhe_amd.utils.synth (splitmix64 -> mod q), so the GPU
box reproduces them without the reference; the fixture stores only SHA-256 digests and a few sample
words of the expected outputs.  No reference source text is stored, only inputs' seeds and outputs.
"""
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from tests.golden import refdriver as rd  # noqa: E402
from liberate_fhe_amd.utils import synth  # noqa: E402

CONFIGS = {
    "small": dict(logN=12, num_scales=5, num_special_primes=2, is_secured=False),
    "bronze": dict(logN=14, num_special_primes=1),
    "silver": dict(logN=15, num_special_primes=2),
    "gold": dict(logN=16, num_special_primes=4),
    "platinum": dict(logN=17, num_special_primes=6),
    # other word widths of the scale primes: 30-bit (fp64 class, far below 2^41) and 45-bit (integer class: >= 2^41)
    "sb30": dict(logN=13, scale_bits=30, num_scales=6, num_special_primes=2, is_secured=False),
    "sb45": dict(logN=13, scale_bits=45, num_scales=6, num_special_primes=2, is_secured=False),
}


def digest(ct):
    """One digest per component over all devices' rows re-ordered to natural prime order."""
    out = []
    for comp in ct.data:
        h = hashlib.sha256()
        for t in comp:
            h.update(np.ascontiguousarray(t.numpy()).tobytes())
        first = comp[0]
        out.append({"sha256": h.hexdigest(), "shape": [list(t.shape) for t in comp],
                    "head": [int(x) for x in first[0, :4]], "tail": [int(x) for x in first[-1, -4:]]})
    return out


def run(name, params, n_dev=1):
    t0 = time.time()
    eng = rd.reference_engine(n_dev, **params)
    rec = {"params": params, "n_devices": n_dev, "q": [int(x) for x in eng.ctx.q], "hash": eng.hash, "ops": {}}
    a, b = synth.ciphertext(eng, 11, 0), synth.ciphertext(eng, 12, 0)
    evk = synth.key_switch_key(eng, 13)
    rotk = synth.key_switch_key(eng, 14, origin="rotation key:7")
    rec["seeds"] = {"ct_a": 11, "ct_b": 12, "evk": 13, "rotk": 14, "rot_delta": 7}
    rec["ops"]["rescale(a)"] = digest(eng.rescale(a))
    prod = eng.cc_mult(a, b, evk)
    rec["ops"]["cc_mult(a,b,evk)"] = digest(prod)
    rec["ops"]["rotate_single(a,rotk)"] = digest(eng.rotate_single(a, rotk))
    rec["ops"]["rotate_single(cc_mult,rotk)"] = digest(eng.rotate_single(prod, rotk))
    rec["ops"]["cc_add(a,b)"] = digest(eng.cc_add(a, b))
    if name not in ("gold", "platinum"):
        rec["ops"]["cc_mult(prod,prod,evk)"] = digest(eng.cc_mult(prod, prod, evk))
    print(f"{name} x{n_dev}: {time.time() - t0:.1f} s", flush=True)
    return rec


def run_levels(params, n_dev, levels):
    """cc_mult (+ relinearize) and rotate_single of level-l ciphertexts for deep levels: rows (and whole devices) drop
    out of the partition as the level grows (rns_partition.py:64-170)."""
    t0 = time.time()
    eng = rd.reference_engine(n_dev, **params)
    rec = {"params": params, "n_devices": n_dev, "q": [int(x) for x in eng.ctx.q], "hash": eng.hash, "ops": {},
           "seeds": {"ct_a": 11, "ct_b": 12, "evk": 13, "rotk": 14, "rot_delta": 7}, "levels": list(levels)}
    evk = synth.key_switch_key(eng, 13)
    rotk = synth.key_switch_key(eng, 14, origin="rotation key:7")
    for level in levels:
        a, b = synth.ciphertext(eng, 11, level), synth.ciphertext(eng, 12, level)
        rec["ops"][f"cc_mult(a,b,evk)@{level}"] = digest(eng.cc_mult(a, b, evk))
        rec["ops"][f"rotate_single(a,rotk)@{level}"] = digest(eng.rotate_single(a, rotk))
        rec["ops"][f"rescale(a)@{level}"] = digest(eng.rescale(a))
        print(f"levels x{n_dev} level {level}: {time.time() - t0:.1f} s", flush=True)
    return rec


def galois_key(eng, seed0):
    """Synthetic galois key: one synthetic rotation key per power-of-two delta."""
    parts = [synth.key_switch_key(eng, seed0 + i, origin=f"rotation key:{d}") for i, d in enumerate(eng.galois_deltas)]
    return type(parts[0])(data=parts, include_special=True, ntt_state=True, montgomery_state=True,
                          origin="galois key", level=0, hash=eng.hash, version=eng.version)


EVALUATOR_OPS = {
    "negate(a)": lambda e, a, b, evk, gk: e.negate(a),
    "mult_int_scalar(a,-7)": lambda e, a, b, evk, gk: e.mult_int_scalar(a, -7),
    "mult(3,a)": lambda e, a, b, evk, gk: e.mult(3, a),
    "mult_scalar(a,0.37)": lambda e, a, b, evk, gk: e.mult_scalar(a, 0.37),
    "add_scalar(a,1.25)": lambda e, a, b, evk, gk: e.add_scalar(a, 1.25),
    "sub(2.5,a)": lambda e, a, b, evk, gk: e.sub(2.5, a),
    "sum(a,gk)": lambda e, a, b, evk, gk: e.sum(a, gk),
    "mean(a,gk)": lambda e, a, b, evk, gk: e.mean(a, gk),
    "pow(a,3,evk)": lambda e, a, b, evk, gk: e.pow(a, 3, evk),
    "square(a,relin=False)": lambda e, a, b, evk, gk: e.square(a, evk, relin=False),
    "var(a,evk,gk)": lambda e, a, b, evk, gk: e.var(a, evk, gk),
    "cov(a,b,evk,gk)": lambda e, a, b, evk, gk: e.cov(a, b, evk, gk),
    "add(a,level_up(b,2))": lambda e, a, b, evk, gk: e.add(a, e.level_up(b, 2)),
    "rescale(a,exact_rounding=False)": lambda e, a, b, evk, gk: e.rescale(a, exact_rounding=False),
}


def run_evaluator(params, n_dev):
    """Digests of the evaluator surface around the hot path on synthetic inputs, from the reference engine."""
    eng = rd.reference_engine(n_dev, **params)
    rec = {"params": params, "n_devices": n_dev, "seeds": {"ct_a": 21, "ct_b": 22, "evk": 23, "gk": 100}, "ops": {}}
    a, b = synth.ciphertext(eng, 21, 0), synth.ciphertext(eng, 22, 0)
    evk, gk = synth.key_switch_key(eng, 23), galois_key(eng, 100)
    for name, fn in EVALUATOR_OPS.items():
        rec["ops"][name] = digest(fn(eng, a, b, evk, gk))
    return rec


# ---- key generation, encrypt / decrypt, encode / decode -------------------------
# Randomness is injected: both engines draw from tests.helpers.SeededCsprng (numpy PCG64) re-seeded at the same
# points, so every integer tensor below is reproducible bit for bit on the GPU box without the reference.
KEYGEN_SEED = 777
PT_SEED = 4242


def reseed(eng, seed):
    from tests.helpers import SeededCsprng
    eng.rng = SeededCsprng(eng.ctx.N, [len(di) for di in eng.ntt.p.d], max(eng.ntt.num_special_primes, 2),
                           devices=list(eng.ntt.devices), seed=seed)


def digest_any(x):
    """Digest of a ciphertext / public key (tuple of tensor lists), a secret key (tensor list) or a key-switch key
    (list of parts, each a (b, a) pair)."""
    if hasattr(x, "data"):
        x = x.data
    if isinstance(x, torch.Tensor):
        return hashlib.sha256(np.ascontiguousarray(x.cpu().numpy()).tobytes()).hexdigest()
    if hasattr(x, "_fields"):
        return digest_any(x.data)
    return [digest_any(y) for y in x]


def integer_plaintext(N, seed=PT_SEED):
    """A small-integer "encoded message" (what encode would hand to encrypt), from splitmix64."""
    return (synth.splitmix64(seed, N) % np.uint64(1 << 30)).astype(np.int64) - (1 << 29)


def keygen_sequence(eng):
    """The operations the fixture records, in order; used by make_golden (reference engine) and by the tests
    (this package's engine) — the random draws line up because the call order is the same."""
    out = {}
    reseed(eng, KEYGEN_SEED)
    sk = eng.create_secret_key()
    out["create_secret_key"] = digest_any(sk)
    pk = eng.create_public_key(sk)
    out["create_public_key(sk)"] = digest_any(pk)
    evk = eng.create_evk(sk)
    out["create_evk(sk)"] = digest_any(evk)
    rotk = eng.create_rotation_key(sk, 5)
    out["create_rotation_key(sk,5)"] = digest_any(rotk)
    conjk = eng.create_conjugation_key(sk)
    out["create_conjugation_key(sk)"] = digest_any(conjk)
    crs = eng.generate_rotation_crs(rotk)
    ksk = eng.create_key_switching_key(sk, sk, a=crs)
    out["create_key_switching_key(sk,sk,a=crs(rotk))"] = digest_any(ksk)
    pt = [torch.from_numpy(integer_plaintext(eng.ctx.N)).to(eng.ntt.devices[d])
          for d in (getattr(eng, "local_ids", None) or range(eng.ntt.num_devices))]
    ct = eng.encrypt(pt, pk)
    out["encrypt(pt,pk)"] = digest_any(ct)
    ct3 = eng.encrypt(pt, pk, level=2)
    out["encrypt(pt,pk,level=2)"] = digest_any(ct3)
    dec = eng.decrypt(ct, sk)
    out["decrypt(ct,sk)"] = digest_any(dec)
    out["decrypt(ct,sk).head"] = [int(v) for v in dec[0].flatten()[:8].cpu()]
    prod = eng.cc_mult(ct, ct, evk)
    out["decrypt(cc_mult(ct,ct,evk),sk)"] = digest_any(eng.decrypt(prod, sk))
    out["decrypt(rotate_single(ct,rotk),sk)"] = digest_any(eng.decrypt(eng.rotate_single(ct, rotk), sk))
    out["decrypt(conjugate(ct,conjk),sk)"] = digest_any(eng.decrypt(eng.conjugate(ct, conjk), sk))
    trip = eng.cc_mult(ct, ct, evk, relin=False)
    out["decrypt(triplet,sk)"] = digest_any(eng.decrypt(trip, sk))
    return out


def run_keygen(params, n_dev):
    eng = rd.reference_engine(n_dev, **params)
    return {"params": params, "n_devices": n_dev, "seed": KEYGEN_SEED, "pt_seed": PT_SEED, "ops": keygen_sequence(eng)}


def message(N, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return g.uniform(-1, 1, N // 2) + 1j * g.uniform(-1, 1, N // 2)


def run_encdec(params, keep):
    """encode (with injected stochastic rounding), decode and the decrypt -> decode tail on a fixed message.
    `keep` = how many leading entries of each vector the fixture stores (all of them for the small ring)."""
    eng = rd.reference_engine(1, **params)
    N = eng.ctx.N
    m = message(N, 31337)
    rec = {"params": params, "message_seed": 31337, "rng_seed": 99, "keep": keep}
    for level in (0, 2):
        reseed(eng, 99)
        pt = eng.encode(m, level=level)[0]
        back = eng.decode([pt], level=level)
        rec[f"encode(level={level})"] = [int(v) for v in pt[:keep]]
        rec[f"decode(encode,level={level})"] = [[float(v.real), float(v.imag)] for v in back[:keep]]
    # decode of an exact integer plaintext: the only fp64 step is the FFT
    ints = integer_plaintext(N)
    dec = eng.decode([torch.from_numpy(ints)], level=0)
    rec["decode(integer_plaintext)"] = [[float(v.real), float(v.imag)] for v in dec[:keep]]
    # encorypt -> decrode with the bias guard (DC term split off and re-added through CRT, eng.py:1472-1681)
    reseed(eng, KEYGEN_SEED)
    sk = eng.create_secret_key()
    pk = eng.create_public_key(sk)
    reseed(eng, 5150)
    ct = eng.encorypt(m + 3.0, pk)
    out = eng.decrode(ct, sk)
    rec["decrode(encorypt(m+3))"] = [[float(v.real), float(v.imag)] for v in out[:keep]]
    rec["decrode_max_err"] = float(np.abs(out - (m + 3.0)).max())
    return rec


def run_bronze_ntt():
    """BASELINE configs[0]: bronze (logN 14), the forward NTT of ONE limb — every limb of the chain, one at a time — through
    the reference's own ntt_context (its tables, its `ntt` / `enter_ntt` methods, special limbs included) with the C oracle
    as the kernel.  Inputs: synth.uniform_rows(seed, ...) lazy words.  One digest per limb."""
    eng = rd.reference_engine(1, **CONFIGS["bronze"])
    ctx, N = eng.ctx, eng.ctx.N
    rows = list(eng.ntt.p.d_special[0])                  # level 0 with the special limbs: every prime of the chain
    rec = {"params": CONFIGS["bronze"], "seed": 2024, "q": [int(ctx.q[i]) for i in rows], "rows": [int(i) for i in rows],
           "ntt": [], "enter_ntt": []}
    x = synth.uniform_rows(2024, rows, ctx.q, N, lazy=True)
    for name in ("ntt", "enter_ntt"):
        t = torch.from_numpy(x.copy())
        getattr(eng.ntt, name)([t], 0, -2)
        out = t.numpy()
        for r in range(len(rows)):
            rec[name].append({"sha256": hashlib.sha256(np.ascontiguousarray(out[r]).tobytes()).hexdigest(),
                              "head": [int(v) for v in out[r, :4]], "tail": [int(v) for v in out[r, -4:]]})
    return rec


W30_PARAMS = dict(buffer_bit_length=30, scale_bits=24, logN=12, num_scales=4, num_special_primes=2, is_secured=False)


def run_w30():
    """The reference's 30-bit / int32 word mode: its ckks_context + ntt_context (constructed through its engine class, the
    only part of the engine that works in this mode) with the C oracle's int32 instantiation as `ntt_cuda`.  Every one of
    the 15 functions once, through the reference's own ntt_context methods, on seeded int32 words: one digest each."""
    eng = rd.reference_engine(1, **W30_PARAMS)
    ctx, ntt = eng.ctx, eng.ntt
    assert ctx.torch_dtype == torch.int32
    N = ctx.N
    rows = list(ntt.p.d_special[0])
    q = [int(ctx.q[i]) for i in rows]
    rng = np.random.default_rng(3030)
    lazy = np.stack([rng.integers(0, 2 * qi, size=N) for qi in q]).astype(np.int32)
    other = np.stack([rng.integers(0, 2 * qi, size=N) for qi in q]).astype(np.int32)
    rec = {"params": W30_PARAMS, "seed": 3030, "q": q, "rows": [int(i) for i in rows], "ops": {}}
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()
    fresh = lambda: torch.from_numpy(lazy.copy())
    for name in ("ntt", "enter_ntt", "intt", "intt_exit", "intt_exit_reduce", "intt_exit_reduce_signed", "mont_redc", "reduce_2q",
                 "make_signed", "make_unsigned", "mont_enter"):
        t = fresh()
        getattr(ntt, name)([t], 0, -2)
        rec["ops"][name] = sha(t)
    a, b = fresh(), torch.from_numpy(other.copy())
    rec["ops"]["mont_mult"] = sha(ntt.mont_mult([a], [b], 0, -2)[0])
    rec["ops"]["mont_add"] = sha(ntt.mont_add([a], [b], 0, -2)[0])
    rec["ops"]["mont_sub"] = sha(ntt.mont_sub([a], [b], 0, -2)[0])
    one = torch.from_numpy(rng.integers(-1, 2, size=N).astype(np.int32))
    rec["ops"]["tile_unsigned"] = sha(ntt.tile_unsigned([one], 0, -2)[0])
    rec["tile_input_sha256"] = hashlib.sha256(np.ascontiguousarray(one.numpy()).tobytes()).hexdigest()
    return rec


def write_pickle_fixture(params):
    """A ciphertext file written by the REFERENCE's save() (host form, eng.py:2001-2015): data, not code."""
    import pickle
    eng = rd.reference_engine(1, **params)
    ct = synth.ciphertext(eng, 31, eng.num_levels - 1)
    host = type(ct)(data=[eng.download_to_cpu.__func__(eng, _as_cuda(c), ct.level, False) for c in ct.data],
                    include_special=False, ntt_state=False, montgomery_state=False, origin=ct.origin, level=ct.level,
                    hash=ct.hash, version=ct.version)
    with open(os.path.join(HERE, "reference_saved_ct.pkl"), "wb") as f:
        pickle.dump(host, f)
    return {"seed": 31, "level": ct.level, "digest": digest(ct), "params": params}


class _Dev:
    type = "cuda"


class _CudaLike(torch.Tensor):
    @property
    def device(self):
        return _Dev()


def _as_cuda(tensors):
    """The reference refuses to download non-CUDA tensors; present the CPU stand-ins as CUDA for that check."""
    out = []
    for t in tensors:
        t = t.clone()
        t.__class__ = _CudaLike
        out.append(t)
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or list(CONFIGS)
    path = os.path.join(HERE, "engine_digests.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    if "keygen" in which:
        which.remove("keygen")
        kpath = os.path.join(HERE, "keygen_encdec.json")
        k = {"keygen_small": run_keygen(CONFIGS["small"], 1), "keygen_small_x2": run_keygen(CONFIGS["small"], 2),
             "keygen_bronze": run_keygen(CONFIGS["bronze"], 1),
             "encdec_small": run_encdec(CONFIGS["small"], 1024), "encdec_silver": run_encdec(CONFIGS["silver"], 128)}
        json.dump(k, open(kpath, "w"), indent=1)
    if "w30" in which:          # the 30-bit / int32 word mode of the ntt_cuda surface
        which.remove("w30")
        json.dump(run_w30(), open(os.path.join(HERE, "w30_ntt.json"), "w"), indent=1)
    if "bronze_ntt" in which:   # BASELINE configs[0]
        which.remove("bronze_ntt")
        json.dump(run_bronze_ntt(), open(os.path.join(HERE, "bronze_ntt.json"), "w"), indent=1)
    if "gold_x8" in which:      # BASELINE configs[3]: the gold chain over 8 devices (11 / 8 rows with the special limbs)
        which.remove("gold_x8")
        out["gold_x8"] = run("gold", CONFIGS["gold"], 8)
        json.dump(out, open(path, "w"), indent=1)
    if "gold_levels" in which:  # gold below level 0, on 1 and on 8 devices (devices run out of rows: 7 alive at 10, 4 at 20, 1 at 32)
        which.remove("gold_levels")
        out["gold_levels"] = run_levels(CONFIGS["gold"], 1, (10, 20, 32))
        json.dump(out, open(path, "w"), indent=1)
        out["gold_levels_x8"] = run_levels(CONFIGS["gold"], 8, (10, 20, 32))
        json.dump(out, open(path, "w"), indent=1)
    for name in which:
        out[name] = run(name, CONFIGS[name])
        if name == "small":
            out["small_x2"] = run(name, CONFIGS[name], 2)
        if name == "small":
            out["evaluator_small"] = run_evaluator(CONFIGS[name], 1)
            out["evaluator_small_x2"] = run_evaluator(CONFIGS[name], 2)
            out["saved_ct"] = write_pickle_fixture(CONFIGS[name])
        json.dump(out, open(path, "w"), indent=1)
