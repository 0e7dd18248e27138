"""Host time per engine op (time to ENQUEUE an op, no synchronisation inside the loop) against its device time:
the margin the one-process-per-GPU sharded path has before the host, not the GPU, paces it."""
import sys, time, warnings
import torch
warnings.filterwarnings("ignore")
sys.path.insert(0, ".")
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

for preset in ("silver", "gold"):
    params = {k: v for k, v in presets.params[preset].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0"], **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for name, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate", lambda: eng.rotate_single(a, rotk))):
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{preset} {name}: host enqueue {1e6 * (t1 - t0) / n:.1f} us/op, device-paced total {1e6 * (t2 - t0) / n:.1f} us/op")
    del eng
    torch.cuda.empty_cache()
