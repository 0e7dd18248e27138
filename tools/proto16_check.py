"""Experiment driver for -DLF_PROTO16: bit-exactness and timing of the 16-words-per-thread tiled pass vs the shipped one."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np, torch
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.utils import synth
ctx = ckks_context(logN=16, num_special_primes=4); ntt = ntt_context(ctx, devices=["cuda:0"])
B = 128; tot = len(ctx.q); N = ctx.N
st = torch.cuda.current_stream().cuda_stream
lo, hi = tot - 30, tot - 5
rows = list(range(lo, hi)); L = len(rows)
x0 = torch.stack([torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, N, lazy=True)) for b in range(8)]).cuda().repeat(B // 8, 1, 1).contiguous()
sl = lambda t: t[0][lo:hi].contiguous()
psi, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
def run(buf, which):
    check(lib.lf_ntt_pass(buf.data_ptr(), B, L, 16, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, which, ql.data_ptr(), qh.data_ptr(),
                          kl.data_ptr(), kh.data_ptr(), 0, st), "pass")
a, b = x0.clone(), x0.clone()
run(a, 2); run(b, 3); torch.cuda.synchronize()
print("bit-exact:", torch.equal(a, b), "mismatches:", int((a != b).sum()))
for which in (2, 3, 2, 3):
    buf = x0.clone()
    for _ in range(25): run(buf, which)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(buf, which)
    e1.record(); torch.cuda.synchronize()
    print(f"which={which}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per launch (25 fp64-class limbs x {B} polys)")
