"""Repeatability soak (development aid):  python tools/soak.py [minutes] [summary-file]
The same headline transform (128 x 30 limbs, logN 16, exact) and the same gold / silver cc_mult and rotate_single are run
over and over for the given wall-clock budget (default 10 minutes, split evenly over the five legs); every result must
equal the first one word for word (a missing barrier or wait shows up as a rare mismatch).  The summary (iterations,
mismatches, rate in the first and in the last tenth of each leg) goes to stdout and, if named, to the summary file."""
import os, sys, time, warnings
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

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
out_path = sys.argv[2] if len(sys.argv) > 2 else None
leg_s = minutes * 60.0 / 5
dev = "cuda:0"
bad = 0
lines = ["# tools/soak.py %.1f min on %s, library digest %s" % (minutes, torch.cuda.get_device_name(0), g.library_digest()[:16]),
         "# leg | iterations | mismatches | rate first tenth (/s) | rate last tenth (/s)"]


def soak(label, run, same, chunk):
    """run() enqueues one iteration, same() compares its result with the first one (synchronises)."""
    global bad
    t0 = time.time()
    marks = []   # (time, iterations) after every checked chunk
    it = miss = 0
    while time.time() - t0 < leg_s:
        for _ in range(chunk):
            run()
            it += 1
        if not same():
            miss += 1
            print(label, "mismatch at iteration", it, flush=True)
        marks.append((time.time() - t0, it))
    span = marks[-1][0]
    head = next(m for m in marks if m[0] >= span / 10)
    tail = next(m for m in marks if m[0] >= span * 0.9)
    r_first = head[1] / head[0]
    r_last = (marks[-1][1] - tail[1]) / max(marks[-1][0] - tail[0], 1e-9) if marks[-1][1] > tail[1] else float("nan")
    bad += miss
    line = "%s | %d | %d | %.1f | %.1f" % (label, it, miss, r_first, r_last)
    print(line, flush=True)
    lines.append(line)


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
x = torch.empty_like(src)


ws = torch.empty((int(lib.lf_ntt_ws_words(B, L, LOGN)),), dtype=torch.int64, device=dev)


def ntt_once():      # through the workspace, as bench.py's step (lf_ntt_ws); the strictly in-place form gives the reference words
    x.copy_(src)
    check(lib.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, ql.data_ptr(),
                        qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st), "lf_ntt_ws")


x.copy_(src)
check(lib.lf_ntt(x.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                 qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, st), "lf_ntt")
first = x.clone()
soak("headline poly-NTT x128 via lf_ntt_ws (every transform compared with lf_ntt's words)", ntt_once, lambda: torch.equal(x, first), 1)
del src, x, first, ntt, ws
torch.cuda.empty_cache()

for name in ("gold", "silver"):
    eng = ckks_engine(**{**presets.params[name], "devices": [dev]})
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for label, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate_single", lambda: eng.rotate_single(a, rotk))):
        ref = fn()
        box = [None]

        def run():
            box[0] = fn()

        def same():   # the last result of the chunk; every 8th op is compared
            return all(torch.equal(box[0].data[c][0], ref.data[c][0]) for c in range(2))

        soak("%s %s (every 8th op compared)" % (name, label), run, same, 8)
    del eng, a, b, evk, rotk
    torch.cuda.empty_cache()
lines.append("SOAK " + ("FAILED" if bad else "OK"))
print(lines[-1])
if out_path:
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
sys.exit(1 if bad else 0)
