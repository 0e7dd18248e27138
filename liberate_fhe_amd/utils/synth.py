"""Deterministic synthetic ciphertexts / keys for benchmarks and parity fixtures.

Words come from splitmix64 (seeded per row) reduced mod q, so any process — this build on the GPU
box, or the reference engine in the build container — regenerates identical tensors from a seed
(SURVEY.md §8c/§8d).  The values are arithmetically valid inputs (canonical residues for
ciphertexts, lazy [0, 2q) NTT/Montgomery words for keys) but encrypt nothing.
"""
from __future__ import annotations

import numpy as np
import torch

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of splitmix64 seeded with `seed` (uint64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + _GAMMA * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform_rows(seed: int, prime_ids, q, N: int, lazy: bool = False) -> np.ndarray:
    """int64 [len(prime_ids), N]; the row of prime index i depends only on (seed, i) — not on which
    device holds it — and is uniform in [0, q_i) (or [0, 2 q_i) when lazy)."""
    out = np.empty((len(prime_ids), N), dtype=np.int64)
    for r, i in enumerate(prime_ids):
        m = np.uint64((2 if lazy else 1) * q[i])
        out[r] = (splitmix64(seed * 1000003 + i * 7919 + 1, N) % m).astype(np.int64)
    return out


def _ids(engine):
    return getattr(engine, "local_ids", None) or list(range(engine.ntt.num_devices))


def ciphertext(engine, seed: int, level: int = 0, ds_type=None):
    """A level-`level` "ciphertext": two polynomials of canonical residues on every local device."""
    ds_type = ds_type or _data_struct(engine)
    q, N = engine.ctx.q, engine.ctx.N
    dest = engine.ntt.p.destination_arrays[level]
    data = []
    for comp in range(2):
        rows = []
        for d in _ids(engine):
            if d < len(dest):
                rows.append(torch.from_numpy(uniform_rows(seed * 2 + comp, dest[d], q, N)).to(engine.ntt.devices[d]))
        data.append(rows)
    return ds_type(data=tuple(data), include_special=False, ntt_state=False, montgomery_state=False,
                   origin="cipher text", level=level, hash=engine.hash, version=engine.version)


def key_switch_key(engine, seed: int, origin: str = "key switch key", ds_type=None):
    """A key-switch key (evk / rotation key layout, eng.py:601-652): per part (b, a), each a list of
    [rows_with_special, N] lazy NTT/Montgomery words.  Stored packed [parts, 2, rows, N] per device."""
    ds_type = ds_type or _data_struct(engine)
    q, N, p = engine.ctx.q, engine.ctx.N, engine.ntt.p
    nparts = p.num_partitions + 1
    ids = _ids(engine)
    packs = []
    for d in ids:
        dest = p.destination_arrays_with_special[0][d]
        pack = np.empty((nparts, 2, len(dest), N), dtype=np.int64)
        for gid in range(nparts):
            for comp in range(2):
                pack[gid, comp] = uniform_rows(seed * 4096 + gid * 2 + comp, dest, q, N, lazy=True)
        packs.append(torch.from_numpy(pack).to(engine.ntt.devices[d]))
    parts = []
    for gid in range(nparts):
        parts.append(ds_type(data=([pk[gid, 0] for pk in packs], [pk[gid, 1] for pk in packs]), include_special=True,
                             ntt_state=True, montgomery_state=True, origin=f"key switch key part index {gid}",
                             level=0, hash=engine.hash, version=engine.version))
    out = ds_type(data=parts, include_special=True, ntt_state=True, montgomery_state=True, origin=origin,
                  level=0, hash=engine.hash, version=engine.version)
    if hasattr(engine, "_remember_pack"):
        engine._remember_pack(out, packs, own=True)   # the parts are views of the packs
    return out


def _data_struct(engine):
    import importlib
    mod = importlib.import_module(type(engine).__module__.rsplit(".", 1)[0] + ".data_struct")
    return mod.data_struct
