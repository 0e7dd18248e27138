import sys, os, warnings
sys.path.insert(0, os.getcwd()); warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
for name in ("silver", "gold"):
    eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    fn = lambda: eng.cc_mult(a, b, evk)
    for warm, n in ((3, 10), (3, 20), (50, 20), (200, 50), (0, 200)):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} warm {warm:3d} timed {n:3d}: {e0.elapsed_time(e1)/n*1e3:.1f} us/op")
    del eng, a, b, evk; torch.cuda.empty_cache()
