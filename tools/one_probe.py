"""silver rotate with the key switch forced to the one-launch / two-pass form: python tools/one_probe.py <min_pairs>"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd._native import lib
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
lib.lf_tune(0, int(sys.argv[1]))
name = sys.argv[2] if len(sys.argv) > 2 else "silver"
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
a = synth.ciphertext(eng, 3, 0)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
b = synth.ciphertext(eng, 4, 0)
evk = synth.key_switch_key(eng, 5)
cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(16)]
for label, fn, per in (("rotate", lambda: eng.rotate_single(a, rotk), 1), ("cc_mult", lambda: eng.cc_mult(a, b, evk), 1),
                       ("rotate batch16", lambda: eng.rotate_single_batch(cts, rotk), 16)):
    for _ in range(60 // per + 2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200 // per
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} min_pairs={sys.argv[1]} {label}: {e0.elapsed_time(e1) / n / per * 1e3:.1f} us per op")
