"""Key format A/B in ONE process (boxes differ by a few percent): the same engine ops with the key read in the planes
format (the default for two-pass ring degrees) and raw, alternating:  python tools/ab_planes.py [gold|silver] [rounds]
Under rocprofv3 (tools/ktrace_cmd.sh planes python3 $PWD/tools/ab_planes.py gold 2) the two inner-product kernels show
up side by side: ks_inner4_kernel (planes) and ks_inner2_kernel (raw)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

name = sys.argv[1] if len(sys.argv) > 1 else "gold"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
engs = {}
for fmt in ("planes", "raw"):
    e = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
    if fmt == "raw":
        e._planes_wanted = lambda: False
    a, b = synth.ciphertext(e, 3, 0), synth.ciphertext(e, 4, 0)
    evk, rotk = synth.key_switch_key(e, 5), synth.key_switch_key(e, 6, origin="rotation key:1")
    cts = [synth.ciphertext(e, 100 + i, 0) for i in range(16)]
    engs[fmt] = (e, a, b, evk, rotk, cts)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n = 60 if name == "gold" else 100
for fmt, (e, a, b, evk, rotk, cts) in engs.items():
    for _ in range(40):
        e.cc_mult(a, b, evk); e.rotate_single(a, rotk)
    e.rotate_single_batch(cts, rotk)
torch.cuda.synchronize()
for r in range(rounds):
    for fmt, (e, a, b, evk, rotk, cts) in engs.items():
        us_m = timed(lambda: e.cc_mult(a, b, evk), n)
        us_r = timed(lambda: e.rotate_single(a, rotk), n)
        us_b = timed(lambda: e.rotate_single_batch(cts, rotk), 3) / 16
        print(f"{name} {fmt:6s} round {r}: cc_mult {us_m:7.1f} us  rotate {us_r:7.1f} us  rotate batch16 {us_b:7.1f} us/ct", flush=True)
