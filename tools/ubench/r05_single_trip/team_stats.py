"""Debug: run the one-launch team transform once and print where its members spend their time (TeamCtl.stats)."""
import ctypes, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np, torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd import _native
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
lib = _native.lib
lib.lf_debug_team_ctl.restype = ctypes.c_void_p
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
lib.lf_tune(4, 1)
for _ in range(3):
    assert lib.lf_ntt(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                      qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0
torch.cuda.synchronize()
ptr = lib.lf_debug_team_ctl()
TEAM_MAX, TS = 256, 16
PAD = 64
nbytes = 4 * TEAM_MAX * PAD + 8 + 8 * TEAM_MAX * TS * 4
buf = (ctypes.c_ubyte * nbytes)()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
assert hip.hipMemcpy(buf, ptr, nbytes, 2) == 0
raw = np.frombuffer(buf, dtype=np.uint8)
words = raw[:4 * TEAM_MAX * PAD].view(np.uint32).reshape(TEAM_MAX, PAD)
arrive, xmask = words[:, 0], words[:, 1]
timeout = raw[4 * TEAM_MAX * PAD:4 * TEAM_MAX * PAD + 4].view(np.uint32)[0]
stats = raw[4 * TEAM_MAX * PAD + 8:].view(np.uint64).reshape(TEAM_MAX, TS, 4)[:64]
print("timeout", timeout, "arrive", arrive[:64].tolist()[:16], "xmask", [hex(v) for v in xmask[:16]])
print("fast teams", int(stats[:, :, 3].min(axis=1).sum()), "of 64")
tick = 0.5e-9   # s_memtime: shader clock, ~2 GHz under load
for name, k in (("column", 0), ("meet", 1), ("tile", 2)):
    v = stats[:, :, k].astype(np.float64) * tick * 1e3
    print(f"{name:7s} ms per member: mean {v.mean():.3f} min {v.min():.3f} max {v.max():.3f}")
