"""A/B of an lf_tune knob in ONE process, alternating settings:  python tools/ab_tune.py <knob> <v1,v2,..> [gold|silver] [rounds]
   knob 1 = LF_TUNE_KS_EXT_COLS_MAX, 2 = LF_TUNE_KS_INNER_BLOCKS (include/ckks_hip.h).  Every setting's results are first
   compared word for word with the first setting's."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd._native import lib
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

knob, values = int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")]
name = sys.argv[3] if len(sys.argv) > 3 else "gold"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
e = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
a, b = synth.ciphertext(e, 3, 0), synth.ciphertext(e, 4, 0)
evk, rotk = synth.key_switch_key(e, 5), synth.key_switch_key(e, 6, origin="rotation key:1")
cts = [synth.ciphertext(e, 100 + i, 0) for i in range(16)]


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n = 60 if name == "gold" else 100
pairs = [(cts[i], cts[(i + 1) % 16]) for i in range(16)]
want = None
for v in values:
    lib.lf_tune(knob, v)
    for _ in range(40):
        e.cc_mult(a, b, evk); e.rotate_single(a, rotk)
    got = [e.cc_mult(a, b, evk), e.rotate_single(a, rotk)] + e.rotate_single_batch(cts, rotk) + e.cc_mult_batch(pairs, evk)
    torch.cuda.synchronize()
    flat = [t.clone() for ct in got for comp in ct.data for t in comp]
    if want is None:
        want = flat
    else:
        assert all(torch.equal(x, y) for x, y in zip(flat, want)), f"knob {knob} = {v}: results differ"
        print(f"knob {knob} = {v}: {len(flat)} result tensors equal to those of {values[0]}", flush=True)
torch.cuda.synchronize()
for r in range(rounds):
    for v in values:
        lib.lf_tune(knob, v)
        us_m = timed(lambda: e.cc_mult(a, b, evk), n)
        us_r = timed(lambda: e.rotate_single(a, rotk), n)
        us_b = timed(lambda: e.rotate_single_batch(cts, rotk), 3) / 16
        us_mb = timed(lambda: e.cc_mult_batch(pairs, evk), 3) / 16
        print(f"{name} knob {knob} = {v:5d}: cc_mult {us_m:7.1f} us  rotate {us_r:7.1f} us  rotate batch16 {us_b:7.1f} us/ct  cc_mult batch16 {us_mb:7.1f} us/ct", flush=True)
