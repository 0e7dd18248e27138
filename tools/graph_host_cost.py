"""Host time of a HIP-graph replay against the eager launches it replaces (bursts of 8 from an idle queue): the native
lf_switch_key of a gold ciphertext (8 launches) and a 2-launch lf_ks_plan_fwd-sized piece, captured with torch.cuda.graph."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

eng = ckks_engine(**{**presets.params["gold"], "devices": ["cuda:0"]})
a = synth.ciphertext(eng, 3, 0)
rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
pinv = pow(5, -1, 2 * eng.ctx.N)
plan, _, fp, ro = eng._op_plan(0, 0)
kp = eng._key_pack(rotk)[0]
out = torch.empty((2, plan.ell, eng.ctx.N), dtype=torch.int64, device="cuda:0")
full = lambda: eng.backend.switch_key_native(plan, a.data[0][0], a.data[1][0], pinv, True, kp, fp, ro, out)
state = eng._ws("ks_state", (plan.ell, eng.ctx.N), 0)
piece = lambda: eng.backend.plan_fwd(plan, state, 0, 3, False)


def host_us(fn, n=8, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ts.append((time.perf_counter() - t0) / n)
        torch.cuda.synchronize()
    ts.sort()
    return 1e6 * ts[len(ts) // 2]


for name, fn in (("lf_switch_key (8 launches)", full), ("lf_ks_plan_fwd of 3 digits (2 launches)", piece)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    g.replay()
    torch.cuda.synchronize()
    print(f"{name}: eager host {host_us(fn):.1f} us, graph replay host {host_us(g.replay):.1f} us")
