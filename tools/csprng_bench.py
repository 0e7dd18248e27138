"""Device-side timing of the sampler kernels at the gold preset's shapes (development aid).

Algorithmic bytes per ChaCha20 state: 128 B state read + 16 B counter write + output
(32 B for the fused samplers = 4 int64 samples, 128 B for raw blocks).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from liberate_fhe_amd.csprng import Csprng


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    N, channels, rep = 1 << 16, 39, 4          # gold: 39 base limbs + 4 special primes on one GPU
    r = Csprng(N, [channels], rep, devices=["cuda:0"])
    q = [[(1 << 40) + 15 + 2 * i for i in range(channels + rep)]]
    L = r.L
    rows = []
    t = timed(lambda: r.randint(q, repeats=rep))
    n = (channels + rep) * L
    rows.append(("randint_fast (43 limbs x 65536)", t, n, n * (128 + 16 + 32)))
    t = timed(lambda: r.discrete_gaussian(non_repeats=[channels], repeats=rep))
    rows.append(("discrete_gaussian_fast (43 x 65536)", t, n, n * (128 + 16 + 32)))
    t = timed(lambda: r.randbytes(repeats=rep))
    rows.append(("chacha20 (43 x 16384 blocks)", t, n, n * (128 + 16 + 128)))
    t = timed(lambda: r.discrete_gaussian(repeats=1))
    rows.append(("discrete_gaussian_fast (1 x 65536, keygen/encrypt shape)", t, L, L * 176))
    c = torch.randn(N, dtype=torch.float64, device="cuda") * 2.0 ** 40
    t = timed(lambda: r.randround(c))
    rows.append(("randround (65536; chacha20 + rounding)", t, N // 16, N // 16 * 272 + N * 24))
    for name, t, states, nbytes in rows:
        print(f"{name:58s} {t * 1e6:9.1f} us  {states * 4 / t / 1e9:8.2f} Gsample/s  {nbytes / t / 1e9:8.1f} GB/s "
              f"({nbytes / t / 8e12 * 100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
