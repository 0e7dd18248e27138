"""rotate_single of a batch of ciphertexts under one key: loop vs rotate_single_batch (development aid)."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
name = sys.argv[1] if len(sys.argv) > 1 else "gold"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(B)]
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
loop = timed(lambda: [eng.rotate_single(ct, rotk) for ct in cts])
batch = timed(lambda: eng.rotate_single_batch(cts, rotk))
evk = synth.key_switch_key(eng, 5)
pairs = [(cts[i], cts[(i + 1) % B]) for i in range(B)]
mloop = timed(lambda: [eng.cc_mult(a, b, evk) for a, b in pairs])
mbatch = timed(lambda: eng.cc_mult_batch(pairs, evk))
print(f"{name} x{B}: cc_mult loop {mloop*1e3/B:.1f} us/op ({B/mloop*1e3:.0f} ops/s)   batch {mbatch*1e3/B:.1f} us/op ({B/mbatch*1e3:.0f} ops/s)")
print(f"{name} x{B}: loop {loop*1e3/B:.1f} us/ct ({B/loop*1e3:.0f} rot/s)   batch {batch*1e3/B:.1f} us/ct ({B/batch*1e3:.0f} rot/s)")
