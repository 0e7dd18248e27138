"""In-process A/B of library builds on the headline step (development aid):
    python tools/ab_inproc.py <lib_a.so|base> <lib_b.so> [...]
Every build is loaded into ONE process and timed on the same buffers in alternation (10 whole transforms, 20 launches
of each pass; 8 rounds), so that box-to-box and minute-to-minute drift (clock throttling: tools/clock_probe.py) cancels.
Prints the median per build and the ratio to the first one."""
import ctypes, os, sys, warnings
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

names = sys.argv[1:] or ["base"]
libs = []
for n in names:
    if n == "base":
        libs.append(_native.lib)
        continue
    L = ctypes.CDLL(os.path.abspath(n))
    for fn in ("lf_ntt", "lf_ntt_pass"):
        getattr(L, fn).argtypes = _native._SIGNATURES[fn]
        getattr(L, fn).restype = ctypes.c_int
    libs.append(L)

LOGN, LIMBS, B = 16, 30, 128
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
rows = list(range(total - LIMBS, total))
x = torch.empty((B, LIMBS, ctx.N), dtype=torch.int64, device=dev)
for b in range(B):
    x[b] = torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)).to(dev)
sl = lambda t: t[0][total - LIMBS:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
st = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)


def full(L):
    assert L.lf_ntt(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                    qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


def one_pass(L, which):
    assert L.lf_ntt_pass(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, which, ql.data_ptr(),
                         qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def timed_passes(L, n):
    """(tiled pass ms, column pass ms) from per-launch events over n transforms issued as their two passes: every launch
    sees the valid output of the other pass (the words between the two passes are in an internal format, PassGeom::f64mid —
    a pass launched twice in a row would read garbage)."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for a, b, c in ev:
        a.record(); one_pass(L, 1); b.record(); one_pass(L, 2); c.record()
    torch.cuda.synchronize()
    return sum(b.elapsed_time(c) for a, b, c in ev) / n, sum(a.elapsed_time(b) for a, b, c in ev) / n


for L in libs:
    for _ in range(5):
        full(L)
torch.cuda.synchronize()
res = {n: {"step": [], "tile": [], "cols": []} for n in names}
for rnd in range(8):
    for n, L in zip(names, libs):
        res[n]["step"].append(timed(lambda: full(L), 10))
        t, c = timed_passes(L, 20)
        res[n]["tile"].append(t)
        res[n]["cols"].append(c)
ref = {k: float(np.median(v)) for k, v in res[names[0]].items()}
for n in names:
    m = {k: float(np.median(v)) for k, v in res[n].items()}
    print(f"{os.path.basename(n):24s} step {m['step']:.4f} ms ({m['step'] / ref['step']:.3f})  tile {m['tile']:.4f} ({m['tile'] / ref['tile']:.3f})"
          f"  cols {m['cols']:.4f} ({m['cols'] / ref['cols']:.3f})   {B / m['step'] * 1e3:8.0f} poly-NTT/s")
