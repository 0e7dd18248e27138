"""Tiled-pass cost per arithmetic class (development aid; run under rocprofv3 --pmc SQ_INSTS_VALU for the counts):
the bench's 30 limbs split into their 25 fp64-class and 5 integer-class rows, each timed alone through lf_ntt_pass."""
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
for name, lo, hi in (("fp64 class (25 limbs)", tot - 30, tot - 5), ("integer class (5 limbs)", tot - 5, tot), ("all 30", tot - 30, tot)):
    rows = list(range(lo, hi)); L = len(rows)
    x = torch.stack([torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, N, lazy=True)) for b in range(8)]).cuda().repeat(B // 8, 1, 1).contiguous()
    sl = lambda t: t[0][lo:hi].contiguous()
    psi, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
    dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
    q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
    for which, label in ((2, "tiled pass"), (1, "column pass")):
        f = lambda: check(lib.lf_ntt_pass(x.data_ptr(), B, L, 16, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, which, ql.data_ptr(), qh.data_ptr(),
                                          kl.data_ptr(), kh.data_ptr(), 0, st), "pass")
        for _ in range(25): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        bf = B * L * (N // 2) * (12 if which == 2 else 4) / 64     # wave-level butterflies per launch
        print(f"{name:26s} {label:11s} {us:8.1f} us/launch  {us * 1e3 / (B * L):7.1f} ns per limb  wave-butterflies {bf:.4g}")
    del x
