"""Repeatability soak (development aid):  python tools/soak.py [iterations]
The same headline transform (128 x 30 limbs, logN 16, exact) and the same gold / silver cc_mult and rotate_single are run
over and over; every result must equal the first one word for word (a missing barrier or wait shows up as a rare mismatch)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.utils import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
bad = 0

LOGN, L, B = 16, 30, 128
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
rows = list(range(total - L, total))
src = torch.empty((B, L, ctx.N), dtype=torch.int64, device=dev)
for b in range(B):
    src[b] = torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)).to(dev)
sl = lambda t: t[0][total - L:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
st = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
first = None
x = torch.empty_like(src)
for it in range(max(10, n // 4)):
    x.copy_(src)
    check(lib.lf_ntt(x.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                     qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st), "lf_ntt")
    if first is None:
        first = x.clone()
    elif not torch.equal(x, first):
        bad += 1
        print("NTT mismatch at iteration", it, int((x != first).sum()))
print("headline transform:", max(10, n // 4), "runs, mismatches", bad)
del src, x, first, ntt
torch.cuda.empty_cache()

for name in ("gold", "silver"):
    eng = ckks_engine(**{**presets.params[name], "devices": [dev]})
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for label, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate_single", lambda: eng.rotate_single(a, rotk))):
        ref = fn()
        miss = 0
        for it in range(n):
            out = fn()
            if not all(torch.equal(out.data[c][0], ref.data[c][0]) for c in range(2)):
                miss += 1
                print(name, label, "mismatch at iteration", it)
        bad += miss
        print(name, label, n, "runs, mismatches", miss)
    del eng, a, b, evk, rotk
    torch.cuda.empty_cache()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
