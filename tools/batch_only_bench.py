"""rotate_single_batch / cc_mult_batch only (for kernel traces): python tools/batch_only_bench.py gold 16 rot|mult"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
name, B, op = sys.argv[1], int(sys.argv[2]), sys.argv[3]
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(B)]
key = synth.key_switch_key(eng, 6, origin="rotation key:1") if op == "rot" else synth.key_switch_key(eng, 5)
pairs = [(cts[i], cts[(i + 1) % B]) for i in range(B)]
fn = (lambda: eng.rotate_single_batch(cts, key)) if op == "rot" else (lambda: eng.cc_mult_batch(pairs, key))
for _ in range(2): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n): fn()
e1.record(); torch.cuda.synchronize()
print(f"{name} {op} x{B}: {e0.elapsed_time(e1)/n*1e3/B:.1f} us per ciphertext")
