"""Does a whole cc_mult capture into a HIP graph, and what does replay cost? (development aid)"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
for name in ("silver", "gold"):
    eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    ref = eng.cc_mult(a, b, evk)
    for _ in range(3): eng.cc_mult(a, b, evk)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): eng.cc_mult(a, b, evk)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = eng.cc_mult(a, b, evk)
    g.replay(); torch.cuda.synchronize()
    ok = all(torch.equal(x, y) for xs, ys in zip(out.data, ref.data) for x, y in zip(xs, ys))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) / n * 1e3
    e0.record()
    for _ in range(n): eng.cc_mult(a, b, evk)
    e1.record(); torch.cuda.synchronize()
    t_eager = e0.elapsed_time(e1) / n * 1e3
    print(f"{name}: graph replay {t_graph:.1f} us/op ({1e6/t_graph:.0f} ops/s), eager {t_eager:.1f} us/op, identical={ok}")
