"""cProfile of the host side of engine ops (which Python / ctypes calls the enqueue time goes to)."""
import cProfile, pstats, sys, warnings
import torch
warnings.filterwarnings("ignore")
sys.path.insert(0, ".")
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

preset, op = sys.argv[1], sys.argv[2]
params = {k: v for k, v in presets.params[preset].items() if k != "devices"}
eng = ckks_engine(devices=["cuda:0"], **params)
a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
evk = synth.key_switch_key(eng, 5)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
fn = (lambda: eng.cc_mult(a, b, evk)) if op == "cc_mult" else (lambda: eng.rotate_single(a, rotk))
for _ in range(40):
    fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fn()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
