"""Shader clock while the headline kernels run (development aid):  python tools/clock_probe.py
A one-wave kernel on a second stream (lf_clock_probe, include/ckks_hip.h) counts core-clock cycles per tick of the
constant 100 MHz counter while the main stream runs (a) nothing, (b) the tiled pass, (c) the column pass, (d) whole
transforms, back to back.  LF_HIP_LIB=<variant .so> probes another build (tools/mkvariant.sh)."""
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

LOGN, L, B = 16, 30, 128
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
rows = list(range(total - L, total))
x = torch.empty((B, L, ctx.N), dtype=torch.int64, device=dev)
for b in range(B):
    x[b] = torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)).to(dev)
sl = lambda t: t[0][total - L:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
main = torch.cuda.current_stream().cuda_stream
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, main)
q_host = np.array([ctx.q[i] for i in rows], dtype=np.int64)
side = torch.cuda.Stream()
SAMPLES, TICKS = 40, 100_000          # 40 samples of 1 ms
out = torch.zeros(2 * SAMPLES, dtype=torch.int64, device=dev)


def one_pass(which):
    check(lib.lf_ntt_pass(x.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, which, ql.data_ptr(),
                          qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, main), "lf_ntt_pass")


def full():
    check(lib.lf_ntt(x.data_ptr(), B, L, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                     qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, main), "lf_ntt")


def measure(name, work, launches):
    torch.cuda.synchronize()
    for _ in range(launches // 4):          # the main stream is busy before the probe starts
        work()
    check(lib.lf_clock_probe(out.data_ptr(), SAMPLES, TICKS, 0, side.cuda_stream), "lf_clock_probe")
    for _ in range(launches):
        work()
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(SAMPLES, 2)
    mhz = o[:, 0] / o[:, 1] * 100.0
    print(f"{name:28s} shader clock MHz: first {mhz[0]:7.1f}  median {np.median(mhz):7.1f}  min {mhz.min():7.1f}  max {mhz.max():7.1f}")


measure("idle", lambda: None, 0)
measure("tiled pass (exact)", lambda: one_pass(2), 60)
measure("column pass", lambda: one_pass(1), 90)
measure("whole transforms", full, 40)
measure("idle again", lambda: None, 0)
