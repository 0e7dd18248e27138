"""Maximum log2(Q) per ring dimension, from the HomomorphicEncryption.org standard tables.

Mirrors the reference's `maximum_qbits` (src/liberate/fhe/context/security_parameters.py:160-201):
piecewise-linear in N through the table points n = 1024 .. 32768 and extended linearly beyond the
last segment (which is how N = 65536 / 131072 get their budgets in the reference).
"""
from __future__ import annotations

RING_DIMS = (1024, 2048, 4096, 8192, 16384, 32768)

# log2(q) budgets, indexed [quantum][distribution][security_bits] -> one value per RING_DIMS entry.
_LOGQ = {
    "pre_quantum": {
        "uniform": {128: (29, 56, 111, 220, 440, 880), 192: (21, 39, 77, 154, 307, 612), 256: (16, 31, 60, 120, 239, 478)},
        "error":   {128: (29, 56, 111, 220, 440, 883), 192: (21, 39, 77, 154, 307, 613), 256: (16, 31, 60, 120, 239, 478)},
        "ternary": {128: (27, 54, 109, 218, 438, 881), 192: (19, 37, 75, 152, 305, 611), 256: (14, 29, 58, 118, 237, 476)},
    },
    "post_quantum": {
        "uniform": {128: (27, 53, 103, 206, 413, 829), 192: (19, 37, 72, 143, 286, 573), 256: (15, 29, 56, 111, 222, 445)},
        "error":   {128: (27, 53, 103, 206, 413, 829), 192: (19, 37, 72, 143, 286, 573), 256: (15, 29, 56, 111, 222, 445)},
        "ternary": {128: (25, 51, 101, 202, 411, 827), 192: (17, 35, 70, 141, 284, 571), 256: (13, 27, 54, 109, 220, 443)},
    },
}


def maximum_qbits(N, security_bits=128, quantum="post_quantum", distribution="uniform") -> float:
    if quantum not in _LOGQ:
        raise AssertionError("Wrong quantum security model!!!")
    if distribution not in ("uniform", "error", "ternary"):
        raise AssertionError("Wrong distribution")
    if security_bits not in (128, 192, 256):
        raise AssertionError("Wrong security level")
    ys = _LOGQ[quantum][distribution][security_bits]
    xs = RING_DIMS
    # segment containing N; end segments extend linearly
    k = 0
    while k < len(xs) - 2 and N > xs[k + 1]:
        k += 1
    x0, x1, y0, y1 = xs[k], xs[k + 1], ys[k], ys[k + 1]
    return y0 + (y1 - y0) * (N - x0) / (x1 - x0)
