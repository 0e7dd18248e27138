"""Upper bound of what a native (one call per group) batched entry could buy: the batched ops captured into a HIP graph
(no Python, no per-launch host cost on replay) against the eager calls (development aid):  python tools/graph_probe_batch.py"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
for name in ("silver", "gold"):
    eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
    evk, rotk = synth.key_switch_key(eng, 5), synth.key_switch_key(eng, 6, origin="rotation key:1")
    cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(16)]
    pairs = [(cts[i], cts[(i + 1) % 16]) for i in range(16)]
    for label, fn in (("rotate_single_batch x16", lambda: eng.rotate_single_batch(cts, rotk)), ("cc_mult_batch x16", lambda: eng.cc_mult_batch(pairs, evk))):
        ref = fn()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = fn()
            g.replay(); torch.cuda.synchronize()
            ok = all(torch.equal(x.data[c][0], y.data[c][0]) for x, y in zip(out, ref) for c in range(2))
            e0.record()
            for _ in range(n): g.replay()
            e1.record(); torch.cuda.synchronize()
            t_graph = e0.elapsed_time(e1) / n * 1e3
        except Exception as e:
            t_graph, ok = float("nan"), f"capture failed: {type(e).__name__}: {e}"[:120]
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        t_eager = e0.elapsed_time(e1) / n * 1e3
        print(f"{name} {label}: graph replay {t_graph:.1f} us, eager {t_eager:.1f} us, identical={ok}", flush=True)
