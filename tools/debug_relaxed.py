import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from liberate_fhe_amd.fhe import ckks_engine, presets
eng = ckks_engine(**{**presets.params["bronze"], "devices": ["cuda:0"]})
d, level = 0, 0
rows, N, logN = eng._rows(d, level, True), eng.ctx.N, eng.ctx.logN
cs = eng._consts(d, level, True)
tw, itw, ninv = eng._tw(d, level, True), eng._tw(d, level, True, True), eng._vec("Ninv", d, level, True)
qv = torch.tensor([eng.ctx.q[i] for i in eng.ntt.p.destination_arrays_with_special[level][d]], device="cuda")
x = torch.stack([torch.randint(0, int(q), (N,), device="cuda", dtype=torch.int64) for q in qv.tolist()])
a, b = x.clone()[None].contiguous(), x.clone()[None].contiguous()
eng.backend.ntt(a, 1, rows, logN, tw, None, cs, relaxed=False)
eng.backend.ntt(b, 1, rows, logN, tw, None, cs, relaxed=True)
print("fwd relaxed vs exact mod q mismatches per row:", ((a[0] % qv[:, None]) != (b[0] % qv[:, None])).sum(dim=1).tolist(), " relaxed canonical:", bool((b[0] < qv[:, None]).all()))
a2, b2 = a.clone(), a.clone()
eng.backend.intt(a2, 1, rows, logN, itw, ninv, 2, cs, relaxed=False)
eng.backend.intt(b2, 1, rows, logN, itw, ninv, 2, cs, relaxed=True)
print("inv relaxed vs exact mismatches per row:", (a2[0] != b2[0]).sum(dim=1).tolist())
