"""Duration of the tiled pass on small grids (development aid; run under tools/ktrace_cmd.sh):
    python tools/tile_floor.py     -> lf_ntt_pass(which = 2) at logN 16 with 1, 2, 8, 30 limbs x 1 polynomial and 30 limbs x 2, 4"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.utils import synth

LOGN = 16
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
st = torch.cuda.current_stream().cuda_stream
for flags in (0, 1):
    for B, L in ((1, 1), (1, 2), (1, 8), (1, -1), (1, -5), (8, -5), (1, 30), (2, 30), (4, 30), (8, 30)):
        if L < 0:                                            # integer-class limbs only (base + special primes)
            L = -L
            lo = total - L
        else:
            lo = total - 5 - L if L < 30 else total - L      # fp64-class limbs only for the small cases
        rows = list(range(lo, lo + L))
        x = torch.stack([torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)) for b in range(B)]).to(dev)
        sl = lambda t: t[0][lo:lo + L]
        psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
        psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
        q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
        def one():
            check(lib.lf_ntt_pass(x.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, flags, 2, ql.data_ptr(),
                                  qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st), "lf_ntt_pass")
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            one()
        e1.record()
        torch.cuda.synchronize()
        print(f"flags {flags}  {B} x {L:2d} limbs = {B * L * 16:5d} tiles: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (back to back)")
