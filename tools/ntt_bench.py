"""Quick device-side timing of the NTT family (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.ntt import ntt_context

def run(logN, nsp, batch, iters=20):
    ctx = ckks_context(logN=logN, num_special_primes=nsp)
    ntt = ntt_context(ctx, devices=["cuda:0"])
    rows = len(ctx.q)
    N = ctx.N
    x = torch.randint(0, 2**40, (batch * rows, N), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    qh_ = np.array(ctx.q, dtype=np.int64)
    from liberate_fhe_amd.ntt import twiddles
    dp = twiddles.dp_pointer(ntt.psi[0], ntt.ql[0], ntt.qh[0], ntt.kl[0], ntt.kh[0], 0, st) if os.environ.get('NO_DP') is None else 0
    args = lambda: (x.data_ptr(), batch, rows, logN, ntt.psi[0].data_ptr(), dp, qh_.ctypes.data if dp else 0, 0, 0, ntt._2q[0].data_ptr(), ntt.ql[0].data_ptr(), ntt.qh[0].data_ptr(), ntt.kl[0].data_ptr(), ntt.kh[0].data_ptr(), 0, st)
    for _ in range(3): check(lib.lf_ntt(*args()), "ntt")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): check(lib.lf_ntt(*args()), "ntt")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    limbs = batch * rows
    bytes_alg = 16 * N * limbs
    print(f"logN={logN} rows={rows} batch={batch}: {ms:.3f} ms/call  {limbs/ms*1e3:.3e} limb-NTT/s  alg {bytes_alg/ms/1e6:.1f} GB/s ({bytes_alg/ms/1e6/8000*100:.1f}% of 8 TB/s)")

if __name__ == "__main__":
    for logN, nsp in ((14, 1), (15, 2), (16, 4)):
        for batch in (1, 8):
            run(logN, nsp, batch)
