"""lf_ntt against lf_ntt_ws (the same transform through a workspace: 6-byte planes between the two passes) in ONE process:
    python tools/ab_ntt_ws.py [logN] [limbs] [batch] [rounds]
First the results are compared word for word — lazy operands, operands of which a sprinkling lies outside [0, 2q) (the third
plane / flag path), and the Montgomery entry (enter_ntt) — then both forms are timed in alternation, whole and per pass."""
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

L = _native.lib
VARIANT = None      # argv[5]: a second build (tools/mkvariant.sh) whose workspace form is timed beside the other two
if len(sys.argv) > 5:
    import ctypes
    VARIANT = ctypes.CDLL(os.path.abspath(sys.argv[5]))
    for fn in ("lf_ntt_ws", "lf_ntt_pass_ws"):
        getattr(VARIANT, fn).argtypes = _native._SIGNATURES[fn]
        getattr(VARIANT, fn).restype = ctypes.c_int
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LIMBS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4 if LOGN >= 16 else 2)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
LIMBS = min(LIMBS, total)
rows = list(range(total - LIMBS, total))
x0 = torch.empty((B, LIMBS, ctx.N), dtype=torch.int64, device=dev)
for b in range(B):
    x0[b] = torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)).to(dev)
sl = lambda t: t[0][total - LIMBS:]
psi, q2, ql, qh, kl, kh, Rs = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh, ntt.Rs))
st = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, st)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
ws = torch.empty((L.lf_ntt_ws_words(B, LIMBS, LOGN),), dtype=torch.int64, device=dev)
ws.fill_(-1)    # contents irrelevant: make them hostile


def plain(x, rs=0):
    assert L.lf_ntt(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, rs, 0, q2.data_ptr(), ql.data_ptr(),
                    qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


def through(x, rs=0, L=L):
    assert L.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, rs, 0, ql.data_ptr(),
                       qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


def one_pass(x, which, w):
    if w:
        assert (VARIANT if w == 2 else L).lf_ntt_pass_ws(x.data_ptr(), ws.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, which,
                                ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0
    else:
        assert L.lf_ntt_pass(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, which, ql.data_ptr(),
                             qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st) == 0


cases = {"lazy operands": x0.clone()}
bad = x0.clone()
gen = torch.Generator(device=dev); gen.manual_seed(7)
idx = torch.randint(0, bad.numel(), (max(64, bad.numel() // 50000),), device=dev, generator=gen)
bad.view(-1)[idx] = torch.randint(-(2 ** 63), 2 ** 63 - 1, (idx.numel(),), dtype=torch.int64, device=dev, generator=gen)
cases["a sprinkling of arbitrary int64 operands"] = bad
for name, src in cases.items():
    for rs_name, rs in (("ntt", 0), ("enter_ntt", Rs.data_ptr())):
        a, b = src.clone(), src.clone()
        plain(a, rs); through(b, rs)
        torch.cuda.synchronize()
        same = torch.equal(a, b)
        if VARIANT is not None:
            b2 = src.clone(); through(b2, rs, VARIANT); torch.cuda.synchronize()
            same = same and torch.equal(a, b2)
        print(f"{rs_name:9s} {name}: {'equal' if same else 'DIFFERENT'}  ({int((a != b).sum())} words differ)", flush=True)
        assert same or os.environ.get("LF_AB_NO_COMPARE") == "1"      # (timing-only experiments with a deliberately wrong variant)

x = x0.clone()


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def timed_passes(w, n):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for a, b, c in ev:
        a.record(); one_pass(x, 1, w); b.record(); one_pass(x, 2, w); c.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b, c in ev) / n, sum(b.elapsed_time(c) for a, b, c in ev) / n


for _ in range(10):
    plain(x); through(x)
torch.cuda.synchronize()
forms = (0, 1) if VARIANT is None else (0, 1, 2)
res = {w: [] for w in forms}
for r in range(ROUNDS):
    for w in forms:
        t = timed((lambda: through(x, 0, VARIANT if w == 2 else L)) if w else (lambda: plain(x)), 20)
        c, p = timed_passes(w, 20)
        res[w].append((t, c, p))
        print(f"round {r} {('in place ', 'workspace', 'ws variant')[w]}: step {t:.4f} ms  column pass {c:.4f}  tiled pass {p:.4f}   {B / t * 1e3:9.0f} poly-NTT/s", flush=True)
med = lambda w, i: float(np.median([v[i] for v in res[w]]))
if VARIANT is not None:
    print(f"variant: step {med(2, 0):.4f} ms ({med(2, 0) / med(1, 0):.4f} x the base workspace form);  column pass {med(2, 1):.4f};  tiled pass {med(2, 2):.4f}")
print(f"median step: in place {med(0, 0):.4f} ms, workspace {med(1, 0):.4f} ms ({med(1, 0) / med(0, 0):.4f} x);  column pass "
      f"{med(0, 1):.4f} -> {med(1, 1):.4f};  tiled pass {med(0, 2):.4f} -> {med(1, 2):.4f}")
