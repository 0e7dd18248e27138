"""Do a slim persistent streaming kernel and a VALU-bound library kernel share the chip?  (development probe)

    python tools/corun_probe.py            (GPU box; builds tools/ubench/corun.so first)

Stream B: tools/ubench/corun.hip — `blocks` workgroups of 256 threads, ALL dispatched at once, streaming `R` bytes with
nontemporal 16-byte loads (or copying them).  Stream A: the library's tiled pass (lf_ntt_pass, which = 2), relaxed over
360 limb rows (the key switch's forward pass at gold, 5 760 tiles) or exact over the headline stack.  For every
configuration: A alone, B alone, both (B launched first, then A; the span ends when both are done).  overlap =
(tA + tB - tAB) / min(tA, tB): 1 = the shorter one is hidden completely, 0 = they ran one after the other."""
import ctypes, os, subprocess, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from liberate_fhe_amd import _native
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context

src = os.path.join(ROOT, "tools", "ubench", "corun.hip")
so = os.path.join(ROOT, "tools", "ubench", "corun.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
co = ctypes.CDLL(so)
P, I, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
co.corun_read.argtypes = [P, U, P, I, I, P]
co.corun_copy.argtypes = [P, P, U, I, I, P]
lib = _native.lib

LOGN, LIMBS = 16, 30
dev = "cuda:0"
ctx = ckks_context(logN=LOGN, num_special_primes=4)
ntt = ntt_context(ctx, devices=[dev])
total = len(ctx.q)
sl = lambda t: t[0][total - LIMBS:]
psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, 0, main.cuda_stream)
q_host = np.array([ctx.q[i] for i in range(total - LIMBS, total)], dtype=np.int64)

big = torch.randint(0, 1 << 40, (64 * 1024 * 1024,), dtype=torch.int64, device=dev)     # 512 MiB to stream
dst = torch.empty_like(big)
sink = torch.zeros(4096, dtype=torch.int64, device=dev)


def tile_pass(x, B, flags):
    assert lib.lf_ntt_pass(x.data_ptr(), B, LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, flags, 2, ql.data_ptr(),
                           qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), 0, main.cuda_stream) == 0


def span(fa, fb, reps=7):
    """median span (ms) of: fb() on the side stream, then fa() on the main stream, until both are done"""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        es = torch.cuda.Event()
        e0.record(main)
        side.wait_event(e0)
        if fb is not None:
            with torch.cuda.stream(side):
                fb()
        if fa is not None:
            fa()
        es.record(side)
        main.wait_event(es)
        e1.record(main)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


print("device", torch.cuda.get_device_name(0), "library", g.library_digest()[:12])
for label, B, flags, n_launch in (("relaxed tiled pass, 12 x 30 limbs (key-switch shape)", 12, 1, 4),
                                   ("exact tiled pass, 128 x 30 limbs (headline)", 128, 0, 1)):
    x = torch.randint(0, 1 << 40, (B, LIMBS, ctx.N), dtype=torch.int64, device=dev)
    fa = lambda: [tile_pass(x, B, flags) for _ in range(n_launch)]
    for _ in range(3):
        fa()
    tA = span(fa, None)
    print(f"\n== A = {n_launch} x {label}: alone {tA * 1e3:.1f} us")
    for kind in ("read", "copy"):
        for blocks in (256, 512, 1024):
            for unroll in (4, 16):
                # size B's work so that alone it lasts about as long as A
                probe_bytes = 256 << 20
                if kind == "read":
                    fb0 = lambda nb=probe_bytes: co.corun_read(big.data_ptr(), nb, sink.data_ptr(), blocks, unroll, side.cuda_stream)
                else:
                    fb0 = lambda nb=probe_bytes: co.corun_copy(big.data_ptr(), dst.data_ptr(), nb, blocks, unroll, side.cuda_stream)
                fb0(); torch.cuda.synchronize()
                t0 = span(None, fb0)
                rate = probe_bytes / (t0 * 1e-3)                     # bytes / s alone (copy: bytes read; moved = 2 x)
                nbytes = int(min(big.numel() * 8, max(16 << 20, rate * tA * 1e-3)) // 4096 * 4096)
                fb = lambda: fb0(nbytes)
                tB = span(None, fb)
                tAB = span(fa, fb)
                ov = (tA + tB - tAB) / min(tA, tB)
                moved = (2 if kind == "copy" else 1) * nbytes / (tB * 1e-3) / 1e12
                print(f"B = {kind:4s} {blocks:5d} blocks x{unroll:2d}: {nbytes / 1e6:7.1f} MB alone {tB * 1e3:7.1f} us ({moved:4.2f} TB/s moved)"
                      f" | both {tAB * 1e3:7.1f} us | sum {(tA + tB) * 1e3:7.1f} | overlap {ov:5.2f}")
    del x
