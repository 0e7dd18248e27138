"""NTT-friendly prime chains for the RNS-CKKS context.

Restates, from the published description, the two prime sequences the reference
draws its RNS moduli from (reference: src/liberate/fhe/context/generate_primes.py:57-96
for the 60-bit message/special primes, :116-203 for the scale-prime sequence,
prim_test.py:4-64 for the primality test).  The reference ships the result as
pickles; this module computes the same lists on demand (a few hundred ms per
(scale_bits, N)) and memoises them — no data files.

* message / special primes of `bits` bits: the NTT primes (q == 1 mod 2N) met
  when walking down from 2^bits - 1.
* scale primes: primes alternating above / below 2^scale_bits, where after each
  pick the search window on the other side is pushed to where the cumulative
  pre-rescale deviation ((2^sb/q_0)^2 ... ) would be cancelled.
"""
from __future__ import annotations

from functools import lru_cache

# Deterministic Miller-Rabin witness set, valid for every n < 3.3e24.
_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41)


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in _MR_BASES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _MR_BASES:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def next_ntt_prime(start: int, M: int, up: bool) -> int:
    """First prime q == 1 (mod M) at or beyond `start` in the given direction."""
    r = (start - 1) % M
    if up:
        q = start if r == 0 else start + (M - r)
        step = M
    else:
        q = start - r
        step = -M
    while not is_prime(q):
        q += step
        if q < 2:
            # walked off the bottom: there is no such prime (the reference's search fails the same way,
            # inside its primality test, once the candidate turns negative — prim_test.py:31)
            raise LookupError(f"no prime = 1 mod {M} below {start}")
    return q


@lru_cache(maxsize=None)
def message_special_primes(bits: int, N: int, how_many: int = 11) -> tuple:
    M = 2 * N
    out = []
    q = (1 << bits) - 1
    while len(out) < how_many:
        q = next_ntt_prime(q, M, up=False)
        out.append(q)
        q -= 2
    return tuple(out)


@lru_cache(maxsize=None)
def scale_primes(scale_bits: int, N: int, how_many: int) -> tuple:
    M = 2 * N
    scale = 1 << scale_bits
    up, down = scale + 1, scale - 1
    first_up = next_ntt_prime(up, M, True)
    first_down = next_ntt_prime(down, M, False)
    # the first pick goes to the side whose nearest candidate is the FARTHER one
    # (generate_primes.py:139-144 of the reference)
    go_up = not ((first_up - scale) < (scale - first_down))
    cumulative = 1
    out = []
    while len(out) < how_many:
        q = next_ntt_prime(up if go_up else down, M, go_up)
        dev = scale / q
        cumulative = cumulative ** 2 * dev ** 2
        if go_up:
            up = q + 2
            target = int((cumulative * scale) // 2 * 2 - 1)
            if target < down:
                down = target
        else:
            down = q - 2
            target = int((cumulative * scale) // 2 * 2 + 1)
            if target > up:
                up = target
        go_up = not go_up
        out.append(q)
    return tuple(out)


def scale_prime_pool(scale_bits: int, N: int) -> tuple:
    """The pool the reference context draws from: 64 primes for logN < 16, else 128; where the ring leaves
    too few NTT primes near 2^scale_bits the request is halved until the search succeeds, and below two
    primes there is no pool (generate_primes.py:234-247, 264-270 of the reference)."""
    logN = N.bit_length() - 1
    how_many = 64 if logN < 16 else 128
    while how_many >= 2:
        try:
            return scale_primes(scale_bits, N, how_many)
        except LookupError:
            how_many //= 2
    raise LookupError(f"not enough NTT primes near 2^{scale_bits} for N = {N}")
