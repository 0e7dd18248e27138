"""Where the host time of a rank of a limb-sharded gold op goes (exchanges stubbed as in tools/host_overhead.py): cProfile
over 400 enqueues of cc_mult / rotate_single on rank 0 of 8."""
import cProfile, pstats, sys, time, warnings
import torch
warnings.filterwarnings("ignore")
sys.path.insert(0, ".")
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
from tools.host_overhead_null import NullComm

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
eng = ckks_engine(devices=["cuda:0"], comm=NullComm(world), **params)
a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
evk = synth.key_switch_key(eng, 5)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
for name, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate", lambda: eng.rotate_single(a, rotk))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    n = 0
    for rep in range(50):
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(8):
            fn()
        pr.disable()
        n += 8
    torch.cuda.synchronize()
    print(f"==== gold {name}, rank 0 of {world}: cumulative host us per op by function (cProfile inflates everything ~1.5x)")
    st = pstats.Stats(pr)
    rows = []
    for (fname, line, func), (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((ct / n * 1e6, tt / n * 1e6, nc / n, f"{fname.split('/')[-1]}:{line}:{func}"))
    rows.sort(reverse=True)
    for ct, tt, nc, nm in rows[:28]:
        print(f"{ct:8.1f} cum {tt:7.1f} self {nc:5.1f} calls  {nm}")
