"""A few headline steps with the one-launch team transform on (argv[1] = 1) or off (0): the program the PMC passes of
tools/pmc_teams.sh profile."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np, torch
from liberate_fhe_amd import _native
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
lib = _native.lib
LOGN, LIMBS, B = 16, 30, 128
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
x = torch.randint(0, 1 << 40, (B, LIMBS, ctx.N), dtype=torch.int64, device=dev)
sl = lambda t: t[0][total - LIMBS:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
st = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in range(total - LIMBS, total)], dtype=np.int64)
lib.lf_tune(4, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
if len(sys.argv) > 2:
    lib.lf_tune(3, int(sys.argv[2]))
for _ in range(6):
    assert lib.lf_ntt(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                      qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0
torch.cuda.synchronize()
