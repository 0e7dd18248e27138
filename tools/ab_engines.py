"""Two engines in ONE process that differ in an engine attribute, alternating (boxes differ by a few percent):
    python tools/ab_engines.py <attribute> <value A> <value B> [gold|silver|bronze] [rounds]
e.g. ks_horner 1 0 — the key switch's extension in Horner form against the sum form.  The attribute is set right after
construction (tables are built on first use).  Results of B are compared with A's, word for word, before anything is timed."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

attr, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
name = sys.argv[4] if len(sys.argv) > 4 else "gold"
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
conv = lambda v: {"0": False, "1": True}.get(v, v)
engs = {}
for tag, v in (("A=" + va, conv(va)), ("B=" + vb, conv(vb))):
    e = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
    assert hasattr(e, attr), attr
    setattr(e, attr, v)
    a, b = synth.ciphertext(e, 3, 0), synth.ciphertext(e, 4, 0)
    evk, rotk = synth.key_switch_key(e, 5), synth.key_switch_key(e, 6, origin="rotation key:1")
    cts = [synth.ciphertext(e, 100 + i, 0) for i in range(16)]
    pairs = [(cts[i], cts[(i + 1) % 16]) for i in range(16)]
    engs[tag] = (e, a, b, evk, rotk, cts, pairs)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n = 60 if name == "gold" else 100
want = None
for tag, (e, a, b, evk, rotk, cts, pairs) in engs.items():
    for _ in range(40):
        e.cc_mult(a, b, evk); e.rotate_single(a, rotk)
    got = [e.cc_mult(a, b, evk), e.rotate_single(a, rotk)] + e.rotate_single_batch(cts, rotk) + e.cc_mult_batch(pairs, evk)
    torch.cuda.synchronize()
    flat = [t.clone() for ct in got for comp in ct.data for t in comp]
    if want is None:
        want = flat
    else:
        assert all(torch.equal(x, y) for x, y in zip(flat, want)), f"{attr}: results differ between {list(engs)}"
        print(f"{attr}: {len(flat)} result tensors of {tag} equal to those of {list(engs)[0]}", flush=True)
for r in range(rounds):
    for tag, (e, a, b, evk, rotk, cts, pairs) in engs.items():
        us_m = timed(lambda: e.cc_mult(a, b, evk), n)
        us_r = timed(lambda: e.rotate_single(a, rotk), n)
        us_b = timed(lambda: e.rotate_single_batch(cts, rotk), 3) / 16
        us_mb = timed(lambda: e.cc_mult_batch(pairs, evk), 3) / 16
        print(f"{name} {attr} {tag:5s} round {r}: cc_mult {us_m:7.1f} us  rotate {us_r:7.1f} us  rotate batch16 {us_b:7.1f} us/ct  cc_mult batch16 {us_mb:7.1f} us/ct", flush=True)
