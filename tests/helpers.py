"""Shared test scaffolding: synthetic limb sets for any logN, seeded inputs, oracle-side constants."""
from __future__ import annotations

import hashlib

import numpy as np

from liberate_fhe_amd.fhe.context import primes as P
from liberate_fhe_amd.fhe.context.ckks_context import bit_reverse_indices, _power_table

R = 1 << 62
LB = (1 << 31) - 1


def primitive_root_2N(q, N):
    """Same search as the context's, but not capped at x < N (tiny test rings need larger x)."""
    e = (q - 1) // (2 * N)
    for x in range(2, 1000):
        g = pow(x, e, q)
        if pow(g, N, q) != 1:
            return g
    raise ValueError(q)


def i64(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.int64))


class Limbs:
    """Montgomery / NTT constants of a list of primes for ring degree 2^logN, as host arrays of the word mode:
    bits = 62 (int64 words, R = 2^62, 31-bit halves) or 30 (the reference's int32 mode: R = 2^30, 15-bit halves)."""

    def __init__(self, logN, q, bits=62):
        self.logN, self.N, self.q = logN, 1 << logN, [int(x) for x in q]
        self.bits, self.R, self.half = bits, 1 << bits, bits // 2
        self.dtype = np.int64 if bits == 62 else np.int32
        N, Rr, lb, h = self.N, self.R, (1 << (bits // 2)) - 1, bits // 2
        arr = lambda v: np.ascontiguousarray(np.asarray(v, dtype=self.dtype))
        self.rows = len(self.q)
        self.k = [(Rr * pow(Rr, -1, qi) - 1) // qi for qi in self.q]
        self.ql, self.qh = arr([x & lb for x in self.q]), arr([x >> h for x in self.q])
        self.kl, self.kh = arr([x & lb for x in self.k]), arr([x >> h for x in self.k])
        self._2q = arr([2 * x for x in self.q])
        self.Rs = arr([Rr * Rr % x for x in self.q])
        self.Ninv = arr([pow(N, -1, x) * Rr % x for x in self.q])
        brev = bit_reverse_indices(logN)
        self.root = [primitive_root_2N(x, N) for x in self.q]
        self.psi_plain = np.stack([_power_table(g, N, x)[brev] for g, x in zip(self.root, self.q)]).astype(self.dtype)
        self.ipsi_plain = np.stack([_power_table(pow(g, -1, x), N, x)[brev] for g, x in zip(self.root, self.q)]).astype(self.dtype)
        self._mont = None

    def mont_tables(self):
        """psi_br / ipsi_br entered into Montgomery form with the ORACLE's mm (as the reference does on device)."""
        if self._mont is None:
            from oracle import oracle as orc
            psi, ipsi = self.psi_plain.copy(), self.ipsi_plain.copy()
            orc.mont_enter(psi, self.Rs, self.rows, self.ql, self.qh, self.kl, self.kh)
            orc.mont_enter(ipsi, self.Rs, self.rows, self.ql, self.qh, self.kl, self.kh)
            self._mont = (psi, ipsi)
        return self._mont

    def mont_args(self):
        return self.ql, self.qh, self.kl, self.kh

    def uniform(self, seed, lazy=False):
        rng = np.random.default_rng(seed)
        return np.stack([rng.integers(0, (2 if lazy else 1) * x, size=self.N, dtype=np.int64) for x in self.q]).astype(self.dtype)


def pick_primes30(logN, n_scale=2, n_message=1, scale_bits=24):
    """NTT-friendly primes of the 30-bit word mode: `n_scale` near 2^scale_bits, `n_message` just below 2^28
    (the reference's message_bits = buffer_bit_length - 2, ckks_context.py:222)."""
    M = 2 << logN
    out, q = [], (1 << scale_bits) + 1
    for _ in range(n_scale):
        q = P.next_ntt_prime(q, M, up=True)
        out.append(q)
        q += 2
    q = (1 << 28) - 1
    for _ in range(n_message):
        q = P.next_ntt_prime(q, M, up=False)
        out.append(q)
        q -= 2
    return out


def pick_primes(logN, n40=2, n60=1):
    """A few NTT-friendly primes for ring degree 2^logN: `n40` near 2^40 and `n60` just below 2^60."""
    M = 2 << logN
    out, q = [], (1 << 40) + 1
    for _ in range(n40):
        q = P.next_ntt_prime(q, M, up=True)
        out.append(q)
        q += 2
    q = (1 << 60) - 1
    for _ in range(n60):
        q = P.next_ntt_prime(q, M, up=False)
        out.append(q)
        q -= 2
    return out


def sha(arr) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


class SeededCsprng:
    """Deterministic stand-in for the engines' Csprng (reference csprng.py:18-323): same method names, shapes and
    value ranges; randomness from numpy's PCG64, so the reference engine in the build container and this package's
    engine on the GPU box draw IDENTICAL tensors from one seed (the reference's own Csprng cannot be seeded).
    `devices`: where the drawn tensors are placed (one entry per logical device)."""

    def __init__(self, N, C, repeats, devices=None, seed=12345, local_ids=None, **_):
        import torch
        self.torch = torch
        self.N, self.C, self.num_repeating_channels = N, list(C), repeats
        self.devices = devices or ["cpu"]
        self.num_devices = len(self.devices)
        self.g = np.random.Generator(np.random.PCG64(seed))

    def _t(self, x, dev=0):
        return self.torch.from_numpy(np.ascontiguousarray(x).astype(np.int64)).to(self.devices[dev])

    def randint(self, amax=3, shift=0, repeats=1):
        # amax scalar -> [repeats, N] shared by every device; amax per-device list of per-row moduli
        # -> [C_dev + repeats, N] with the trailing `repeats` rows identical on every device.
        if not isinstance(amax, (list, tuple)):
            x = self.g.integers(0, amax, size=(max(repeats, 1), self.N)) + shift
            return [self._t(x, d) for d in range(self.num_devices)]
        out = []
        rep_rows = None
        for dev, q in enumerate(amax):
            q = list(q)
            n_rep = repeats
            body = q[: len(q) - n_rep] if n_rep else q
            rows = [self.g.integers(0, qi, size=self.N) + shift for qi in body]
            if n_rep:
                if rep_rows is None:
                    rep_rows = [self.g.integers(0, qi, size=self.N) + shift for qi in q[len(q) - n_rep:]]
                rows += rep_rows
            out.append(self._t(np.stack(rows), dev))
        return out

    def discrete_gaussian(self, non_repeats=0, repeats=1, sigma=3.2):
        x = np.rint(self.g.normal(0.0, 3.2, size=(max(repeats, 1), self.N)))
        return [self._t(x, d) for d in range(self.num_devices)]

    def randround(self, coef):
        dev = coef.device if isinstance(coef, self.torch.Tensor) else "cpu"
        c = coef.cpu().numpy() if isinstance(coef, self.torch.Tensor) else np.asarray(coef)
        fl = np.floor(c)
        r = fl + (self.g.random(c.shape) < (c - fl))
        return self.torch.from_numpy(r.astype(np.int64)).to(dev)
