"""CKKS encode / decode (fp64 negacyclic FFT) and the Galois exponent of a rotation.

API mirror of the reference's src/liberate/fhe/encdec/encdec.py (`encode`, `decode`, `rotate`,
`conjugate`): same slot order (so plaintexts / ciphertexts are interchangeable with the reference's)
and the same fp64 pipeline, which stays on torch.fft — the reference does the same and SURVEY.md §8(f)
ranks it as a "next" row; it is not part of the integer hot path.

Slot order.  The reference orders slots so that the ring automorphism X -> X^3 rotates the decoded
vector by one.  It derives that order by matching cycles of two permutations (encdec.py:65-121,
199-209): the "circular shift" of both halves of [0, N) and the action i -> 3i + 1 (mod N) of the
automorphism on the odd exponents 2i + 1.  `_slot_permutation` below builds the same map directly:
walk each orbit of i -> 3i+1 starting at the image of its smallest member, and lay the orbit along
the corresponding shift cycle.
"""
from __future__ import annotations

import numpy as np
import torch

_perm_cache = {}
_twist_cache = {}


def _orbits(perm: np.ndarray):
    """Cycles of `perm`, each listed from perm[s] round to s, s = smallest unvisited element."""
    n = len(perm)
    seen = np.zeros(n, dtype=bool)
    out = []
    for s in range(n):
        if seen[s]:
            continue
        cyc = []
        x = int(perm[s])
        while True:
            cyc.append(x)
            seen[x] = True
            if x == s:
                break
            x = int(perm[x])
        out.append(cyc)
    return out


def _slot_permutation(N: int):
    """(pre_perm [N/2], post_perm [N]) as numpy int64 — the reference's `prepost_perms`."""
    half = N // 2
    shift = np.concatenate([np.roll(np.arange(half), 1), np.roll(np.arange(half), -1) + half])
    fold = (3 * np.arange(N) + 1) % N
    a, b = _orbits(shift), _orbits(fold)
    assert [len(c) for c in a] == [len(c) for c in b]
    post = np.zeros(N, dtype=np.int64)
    post[np.concatenate([np.array(c) for c in b])] = np.concatenate([np.array(c) for c in a])
    pre = np.argsort(post)[:half]
    return pre, post


def prepost_perms(N, device="cuda:0"):
    key = (N, str(device))
    if key not in _perm_cache:
        pre, post = _slot_permutation(N)
        _perm_cache[key] = (torch.from_numpy(pre).to(device), torch.from_numpy(post).to(device))
    return _perm_cache[key]


def _twist(N, device, sign):
    key = (N, str(device), sign)
    if key not in _twist_cache:
        ang = sign * 1j * torch.pi * torch.arange(N, device=device, dtype=torch.float64) / N
        _twist_cache[key] = torch.exp(ang)
    return _twist_cache[key]


def generate_twister(N, device="cuda:0"):
    return _twist(N, device, -1)


def generate_skewer(N, device="cuda:0"):
    return _twist(N, device, +1)


def encode(m, rng=None, scale=2 ** 40, deviation=1.0, device="cuda:0", norm="forward",
           return_without_scaling=False):
    """Message (N/2 complex slots) -> N real polynomial coefficients (encdec.py:273-298)."""
    N = len(m) * 2
    pre, _ = prepost_perms(N, device)
    mm = torch.from_numpy(np.array(m * deviation)).to(device)
    spread = torch.zeros((N,), dtype=mm.dtype, device=mm.device)
    spread[pre] = mm
    spread = spread + spread.conj().flip(0)
    coeffs = (torch.fft.fft(spread, norm=norm) * generate_twister(N, device)).real
    if return_without_scaling:
        return coeffs
    return rng.randround(coeffs * np.float64(scale))


def decode(m, scale=2 ** 40, correction=1.0, norm="forward", return_without_scaling=False):
    """N polynomial coefficients -> N complex values, the message in the first N/2 (encdec.py:301-323)."""
    N = len(m)
    device = m.device
    _, post = prepost_perms(N, device)
    vals = torch.fft.ifft(m * generate_skewer(N, device), norm=norm)
    if not return_without_scaling:
        vals = vals / scale * correction
    out = torch.zeros_like(vals)
    out[post] = vals
    return out


def galois_exponent(N: int, delta: int) -> int:
    """p such that rotating the slots by `delta` is X -> X^p: p = 3^(delta mod N) mod 2N (encdec.py:224-229)."""
    return pow(3, delta % N, 2 * N)


def conjugation_exponent(N: int) -> int:
    """Complex conjugation of the slots is X -> X^(2N-1) (encdec.py:249-253)."""
    return 2 * N - 1
