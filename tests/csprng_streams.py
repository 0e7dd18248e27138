"""Inputs of the sampler stream fixtures, shared by the generator (tests/golden/make_golden_csprng.py, which
replays them through the REFERENCE's Csprng class) and by the tests (which replay them through the product)."""
import hashlib

import numpy as np

STREAM_CONFIGS = {
    "small": dict(num_coefs=1024, num_channels=[3, 2], num_repeating_channels=2, n_dev=2),
    "gold_like": dict(num_coefs=65536, num_channels=[10, 10, 10, 9], num_repeating_channels=4, n_dev=4),
}
STREAM_KEY = [0x03020100, 0x07060504, 0x0B0A0908, 0x0F0E0D0C, 0x13121110, 0x17161514, 0x1B1A1918, 0x1F1E1D1C]
STREAM_NONCE = [0x4A000000, 0x00000009]
STREAM_Q = [1099511627689, 1152921504606846883, 97, (1 << 61) - 1, 12345678901, 1073741827, 2, 3, 1125899906842597,
            576460752303423433]


def stream_calls(cfg):
    """The call sequence both sides replay: (name, method, kwargs).  Moduli lists are cut from STREAM_Q."""
    shares, rep = cfg["num_channels"], cfg["num_repeating_channels"]
    q_lists = [[STREAM_Q[(d + i) % len(STREAM_Q)] for i in range(max(s - 1, 1) + rep)] for d, s in enumerate(shares)]
    return [
        ("ternary", "randint", dict(amax=3, shift=-1, repeats=1)),
        ("error", "discrete_gaussian", dict(repeats=1)),
        ("uniform", "randint", dict(amax=q_lists, repeats=rep)),
        ("error2", "discrete_gaussian", dict(non_repeats=[1] * len(shares), repeats=2)),
        ("bytes", "randbytes", dict()),
        ("binary", "randint", dict(amax=2, shift=0, repeats=1)),
        ("uniform_again", "randint", dict(amax=q_lists, repeats=rep)),
    ]


def randround_input(n):
    g = np.random.Generator(np.random.PCG64(99))
    c = g.normal(0.0, 2.0 ** 40, n)
    c[:8] = [0.0, -0.0, 0.5, -0.5, 1.5, -2.5, 3.0, -7.0]
    c[8:12] = [2.0 ** 52 + 1, -(2.0 ** 52 + 1), 1e-12, -1e-12]
    return c


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()
