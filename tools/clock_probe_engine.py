"""Shader clock while engine ops run (development aid):  python tools/clock_probe_engine.py [gold|silver|bronze]
lf_clock_probe on a second stream beside loops of cc_mult / rotate_single (see tools/clock_probe.py)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

name = sys.argv[1] if len(sys.argv) > 1 else "gold"
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
evk = synth.key_switch_key(eng, 5)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
side = torch.cuda.Stream()
SAMPLES, TICKS = 30, 100_000
out = torch.zeros(2 * SAMPLES, dtype=torch.int64, device="cuda:0")


def measure(label, fn, n):
    torch.cuda.synchronize()
    for _ in range(n // 4):
        fn()
    check(lib.lf_clock_probe(out.data_ptr(), SAMPLES, TICKS, 0, side.cuda_stream), "lf_clock_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(SAMPLES, 2)
    mhz = o[:, 0] / o[:, 1] * 100.0
    us = e0.elapsed_time(e1) / max(n, 1) * 1e3
    print(f"{name} {label:14s} {us:8.1f} us/op   shader clock MHz: median {np.median(mhz):7.1f}  min {mhz.min():7.1f}  max {mhz.max():7.1f}")


per = {"gold": 90, "silver": 260, "bronze": 400}[name]
measure("idle", lambda: None, 0)
measure("cc_mult", lambda: eng.cc_mult(a, b, evk), per)
measure("rotate_single", lambda: eng.rotate_single(a, rotk), per)
