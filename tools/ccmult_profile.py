"""Kernel-level timing of one engine op (development aid): python tools/ccmult_profile.py gold cc_mult [--mark] [--tune=<knob>:<value>]

--mark: bracket the timed loop with two tiny marker kernels (torch fill of a 1-element tensor with 12345 / 54321 is
not distinguishable in a trace, so the markers are lf_reduce_2q launches on a [1, 2] tensor — `ew_kernel<...>` with
grid 1), so that tools/summarize_engine_ops.py can separate steady-state dispatches from set-up."""
import sys, os, warnings
sys.path.insert(0, os.environ.get("LF_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # LF_PKG_ROOT: A/B another checkout
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
name = sys.argv[1] if len(sys.argv) > 1 else "gold"
op = sys.argv[2] if len(sys.argv) > 2 else "cc_mult"
mark = "--mark" in sys.argv
for arg in sys.argv:                       # --tune=<knob>:<value> (lf_tune, include/ckks_hip.h) before anything runs
    if arg.startswith("--tune="):
        from liberate_fhe_amd._native import lib as _lib
        _k, _v = arg[7:].split(":")
        _lib.lf_tune(int(_k), int(_v))
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
evk = synth.key_switch_key(eng, 5)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
if op in ("rotate_batch", "cc_mult_batch"):     # 16 ciphertexts under one key: 4 groups of 4 per key-switch launch set, two lanes
    cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(16)]
    pairs = [(cts[i], cts[(i + 1) % 16]) for i in range(16)]
    fn = (lambda: eng.rotate_single_batch(cts, rotk)) if op == "rotate_batch" else (lambda: eng.cc_mult_batch(pairs, evk))
else:
    fn = (lambda: eng.cc_mult(a, b, evk)) if op == "cc_mult" else (lambda: eng.rotate_single(a, rotk))
for _ in range(5 if op.endswith('_batch') else 40): fn()   # steady-state clocks (tools/warm_probe.py)
torch.cuda.synchronize()


def marker():
    from liberate_fhe_amd._native import lib
    t = torch.zeros((1, 2), dtype=torch.int64, device="cuda:0")
    q2 = torch.full((1,), 10, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    lib.lf_make_signed(t.data_ptr(), 1, 2, q2.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 3 if op.endswith('_batch') else 10
if mark: marker()
e0.record()
for _ in range(n): fn()
e1.record(); torch.cuda.synchronize()
if mark: marker()
per = 16 if op.endswith("_batch") else 1
print(f"{name} {op}: {e0.elapsed_time(e1)/n/per*1e3:.1f} us/op  {n*per/e0.elapsed_time(e1)*1e3:.1f} ops/s")
