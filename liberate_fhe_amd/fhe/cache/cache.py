"""Prime-table cache files in the reference's format (src/liberate/fhe/cache/cache.py:14-32,
context/generate_primes.py:17-96, 250-285).

This build computes its prime chains on demand (fhe/context/primes.py) and needs no files; the two
functions exist so that a deployment which shares a cache folder with the reference can (re)create
or clear it: `generate_cache` writes logN_N_M.pkl, message_special_primes.pkl and scale_primes.pkl
with exactly the contents the reference generates (checked against its shipped tables in
tests/test_generate_engine_cpu.py).
"""
from __future__ import annotations

import glob
import os
import pickle

from ..context import primes

path_cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources")

LOGN = list(range(12, 18))
MESSAGE_BITS = (28, 60)
SCALE_BITS = range(20, 60)


def clean_cache(path=None):
    for file in glob.glob(os.path.join(path or path_cache, "*.pkl")):
        try:
            os.unlink(file)
        except Exception as e:  # same forgiving behaviour as the reference
            print(e)


def tables():
    N = [2 ** x for x in LOGN]
    logN_N_M = {"logN": list(LOGN), "N": N, "M": [2 * n for n in N]}
    message = {mb: {n: list(primes.message_special_primes(mb, n)) for n in N} for mb in MESSAGE_BITS}
    scale = {}
    for n in N:
        for sb in SCALE_BITS:
            try:
                scale[(sb, n)] = list(primes.scale_prime_pool(sb, n))
            except LookupError:
                scale[(sb, n)] = f"ERROR!!! sb = {sb}, N = {n}. Not enough primes."
    return {"logN_N_M.pkl": logN_N_M, "message_special_primes.pkl": message, "scale_primes.pkl": scale}


def generate_cache(path=None):
    path = path or path_cache
    os.makedirs(path, exist_ok=True)
    for name, table in tables().items():
        target = os.path.join(path, name)
        if not os.path.exists(target):
            with open(target, "wb") as f:
                pickle.dump(table, f)
