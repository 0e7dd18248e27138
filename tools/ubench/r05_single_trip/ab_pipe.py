"""In-process A/B of the chunk pipeline of lf_ntt (lf_tune LF_TUNE_NTT_PIPE_CHUNKS / _BLOCKS) on the headline step:
128 polynomials x 30 limbs, logN 16.  Every setting is first checked word for word against the unpipelined transform of the
same input, then timed in alternation with it (rounds of 10 steps)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd import _native
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.utils import synth
lib = _native.lib
CHUNKS, BLOCKS, TEAMS = 2, 3, 4

LOGN, LIMBS, B = 16, 30, int(os.environ.get("B", "128"))
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
rows = list(range(total - LIMBS, total))
src = torch.empty((B, LIMBS, ctx.N), dtype=torch.int64, device=dev)
for b in range(B):
    src[b] = torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)).to(dev)
x = src.clone()
sl = lambda t: t[0][total - LIMBS:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
st = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)


def full():
    assert lib.lf_ntt(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                      qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        full()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def setting(chunks, blocks):
    # chunks = -1: the one-launch XCD-team form (LF_TUNE_NTT_TEAMS) instead of the chunk pipeline
    lib.lf_tune(TEAMS, -chunks if chunks < 0 else 0)   # -1 .. -7: team form, access-flavour variant
    lib.lf_tune(CHUNKS, max(chunks, 0))
    lib.lf_tune(BLOCKS, blocks)


setting(0, 1024)
full(); torch.cuda.synchronize()
want = x.clone()
for _ in range(5):
    full()
torch.cuda.synchronize()
grid = [(c, b) for c in (2, 4, 8, 16, 32) for b in (256, 512, 1024, 2048)]
if len(sys.argv) > 1:
    grid = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
res = {}
for c, b in grid:
    setting(c, b)
    x.copy_(src)
    full(); torch.cuda.synchronize()
    ok = bool(torch.equal(x, want))
    ts_base, ts = [], []
    for _ in range(5):
        setting(0, 1024); ts_base.append(timed(10))
        setting(c, b); ts.append(timed(10))
    ts_base.sort(); ts.sort()
    res[(c, b)] = (ok, ts_base[2], ts[2])
    print(f"chunks {c:2d} blocks {b:5d}: parity {'ok' if ok else 'MISMATCH'} | base {ts_base[2]:.4f} ms | piped {ts[2]:.4f} ms | "
          f"{B / ts[2]:.1f} k poly-NTT/s vs {B / ts_base[2]:.1f} k ({ts[2] / ts_base[2] - 1:+.1%})", flush=True)
