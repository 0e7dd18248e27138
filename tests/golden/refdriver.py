"""Import the reference's Python layers in the BUILD container and drive them on CPU.

Only used by tests/golden/make_golden.py (fixture generation) and by tests that are skipped when
/root/reference is absent (it never exists on the GPU box).  The reference's five CUDA extension
modules cannot be built here; this module registers stand-ins for them:

  * liberate.ntt.ntt_cuda                -> the C oracle (oracle/ckks_oracle.c) behind the
                                            reference's exact 15-function list-of-tensors signature
  * liberate.csprng.*_cuda               -> NumPy samplers (deterministic, seeded) — the reference's
                                            CSPRNG is unseedable, so fixtures inject their own randomness

plus the small environment shims listed in SURVEY.md §8(c): np.bool8 alias, a writable copy of the
shipped prime pickles, no-op pin_memory, Tensor.cuda -> clone, and a CPU-safe `decode`.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REFERENCE_SRC = "/root/reference/src"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "liberate"))


def _make_ntt_cuda_standin():
    """`ntt_cuda` stand-in: reference signatures (ntt.cpp:8-113, 421-437), oracle arithmetic."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from tests.oracle_backend import make_ops
    return make_ops("liberate.ntt.ntt_cuda")


from tests.helpers import SeededCsprng  # noqa: E402  (portable: the GPU-box tests use the same class)


_state = {}


def load_reference(seed=12345):
    """Returns the reference's `liberate.fhe` module with the stand-ins installed."""
    if "fhe" in _state:
        return _state["fhe"]
    if not reference_available():
        raise RuntimeError("reference not present")
    if not hasattr(np, "bool8"):
        np.bool8 = np.bool_
    sys.path.insert(0, REFERENCE_SRC)
    sys.modules["liberate.ntt.ntt_cuda"] = _make_ntt_cuda_standin()
    from tests.oracle_csprng import make_modules
    for name, mod in make_modules().items():
        sys.modules["liberate.csprng." + name] = mod        # oracle-backed sampler kernels (CPU)
    import liberate.csprng.csprng as csprng_mod  # noqa: E402
    import liberate.csprng as csprng_pkg  # noqa: E402
    _state["RefCsprng"] = csprng_mod.Csprng                # the reference's own class, kept for the sampler tests
    csprng_mod.Csprng = SeededCsprng
    csprng_pkg.Csprng = SeededCsprng
    from liberate import fhe  # noqa: E402
    # `liberate.fhe.ckks_engine` the attribute is the CLASS (fhe/__init__ re-exports it); the module
    # object has to come from sys.modules.
    eng_mod = sys.modules["liberate.fhe.ckks_engine"]
    eng_mod.Csprng = SeededCsprng

    # CPU plumbing
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()

    encdec_mod = sys.modules["liberate.fhe.encdec.encdec"]
    orig_decode = encdec_mod.decode

    def decode_cpu(m, scale=2 ** 40, correction=1.0, norm="forward", return_without_scaling=False):
        N = len(m)
        device = "cpu"
        if (N, device) not in encdec_mod.perm_cache:
            encdec_mod.perm_cache[(N, device)] = encdec_mod.prepost_perms(N, device=device)
        pre_perm, post_perm = encdec_mod.perm_cache[(N, device)]
        if (N, device) not in encdec_mod.skewer_cache:
            encdec_mod.skewer_cache[N, device] = encdec_mod.generate_skewer(N, device)
        skewer = encdec_mod.skewer_cache[N, device]
        mm = encdec_mod.poly2m(m, skewer, norm=norm)
        if not return_without_scaling:
            mm = mm / scale * correction
        return encdec_mod.post_permute(mm, post_perm)

    encdec_mod.decode = decode_cpu
    eng_mod.decode = decode_cpu

    cache = tempfile.mkdtemp(prefix="lfa_refcache_")
    res = os.path.join(REFERENCE_SRC, "liberate/fhe/cache/resources")
    for f in os.listdir(res):
        shutil.copy(os.path.join(res, f), cache)
    _state.update(fhe=fhe, cache=cache, eng_mod=eng_mod)
    return fhe


def reference_engine(n_devices=1, **params):
    fhe = load_reference()
    params = dict(params)
    params.pop("devices", None)
    return fhe.ckks_engine(devices=["cpu"] * n_devices, cache_folder=_state["cache"],
                           read_cache=False, save_cache=False, **params)


def reference_context(**params):
    load_reference()
    from liberate.fhe.context.ckks_context import ckks_context
    return ckks_context(cache_folder=_state["cache"], read_cache=False, save_cache=False, **params)


def reference_csprng(num_coefs, num_channels, num_repeating_channels, n_dev, key, nonce):
    """The REFERENCE's Csprng class (csprng.py) on CPU with the oracle's sampler kernels underneath and a
    caller-chosen key / nonce (the reference always draws them from os.urandom, so they are set after
    construction through its own initialize_states)."""
    load_reference()
    r = _state["RefCsprng"](num_coefs, list(num_channels), num_repeating_channels, devices=["cpu"] * n_dev)
    # The reference hands out the address of a temporary (discrete_gaussian_sampler.py:111-113: `btree_conti`
    # is local, so `btree_ptr` dangles once the builder returns).  Re-flatten the reference's own table into
    # an array that stays alive.
    r._btree_flat = np.ascontiguousarray(r.btree.T.ravel(), dtype=np.uint64)
    r.btree_ptr = r._btree_flat.__array_interface__["data"][0]
    r.key = [torch.tensor(key, dtype=torch.int64) for _ in range(n_dev)]
    r.nonce = [torch.tensor(nonce, dtype=torch.int64) for _ in range(n_dev)]
    for d in range(n_dev):
        r.initialize_states(d)
    return r
