"""ctypes front-end of the CPU oracle (oracle/ckks_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of ckks_oracle.c.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package never does.

All functions take C-contiguous numpy arrays (int64 data / constants, int32 index tables) and
mutate `a` in place exactly where the reference op does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ckks_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libckks_oracle.so")


# Two builds of the same C text.  "portable" (-O2, no -march): what the checker uses everywhere — it is prebuilt in the build
# container and must run on whatever host the GPU box has.  "native" (-O3 -march=native): the fair CPU port for bench.py's
# cpu_baseline — compiled ON the box that times it (never shipped: another CPU could fault on its instructions), -fwrapv kept
# (the reference's signed arithmetic wraps), same words as the portable build (tests/test_oracle_cpu.py).
_FLAGS = {"portable": ["-O2"], "native": ["-O3", "-march=native", "-funroll-loops"]}


def _lib_path(flavor):
    if flavor == "portable":
        return _LIB
    import hashlib
    import platform
    try:
        cpu = next((ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")), "")
    except OSError:
        cpu = ""
    tag = hashlib.sha256((platform.machine() + cpu).encode()).hexdigest()[:12]
    return os.path.join(_HERE, "_build", f"libckks_oracle_native_{tag}.so")


def build(force: bool = False, flavor: str = "portable") -> str:
    path = _lib_path(flavor)
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(os.path.join(_HERE, "ckks_oracle_impl.h")))
    if force or not os.path.exists(path) or os.path.getmtime(path) < newest:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.check_call(["gcc"] + _FLAGS[flavor] + ["-fwrapv", "-fopenmp", "-shared", "-fPIC", "-o", path, _SRC])
    return path


_lib = None
_libs = {}
_flavor = "portable"


def _load(flavor):
    got = _libs.get(flavor)
    if got is None:
        got = ctypes.CDLL(build(flavor=flavor))
        i64 = ctypes.c_int64
        got.lfo_mm_scalar.restype = i64
        got.lfo_mm_scalar.argtypes = [i64] * 6
        got.lfo_redc_scalar.restype = i64
        got.lfo_redc_scalar.argtypes = [i64] * 5
        i32 = ctypes.c_int32
        got.lfo30_mm_scalar.restype = i32
        got.lfo30_mm_scalar.argtypes = [i32] * 6
        got.lfo30_redc_scalar.restype = i32
        got.lfo30_redc_scalar.argtypes = [i32] * 5
        _libs[flavor] = got
    return got


def use_build(flavor: str) -> str:
    """Select the build every call below goes to ("portable" | "native"); returns the previous selection."""
    global _lib, _flavor
    if flavor not in _FLAGS:
        raise ValueError(flavor)
    before, _flavor = _flavor, flavor
    _lib = _load(flavor)
    return before


def build_flags(flavor=None):
    return "gcc " + " ".join(_FLAGS[flavor or _flavor] + ["-fwrapv", "-fopenmp"])


def lib():
    global _lib
    if _lib is None:
        _lib = _load(_flavor)
    return _lib


def _p(x: np.ndarray):
    assert x.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return ctypes.c_void_p(x.ctypes.data)


def _i64(x):
    assert x.dtype == np.int64, x.dtype
    return _p(x)


def _i32(x):
    assert x.dtype == np.int32, x.dtype
    return _p(x)


def _mode(a):
    """(symbol prefix, word dtype, C type of N) of the word mode an array belongs to: int64 = the reference's 62-bit mode
    (lfo_*), int32 = its 30-bit mode (lfo30_*: the same C text compiled over int32 words, ckks_oracle_impl.h)."""
    if a.dtype == np.int64:
        return "lfo_", np.int64, ctypes.c_int64
    if a.dtype == np.int32:
        return "lfo30_", np.int32, ctypes.c_int32
    raise TypeError(f"oracle arrays are int64 (62-bit mode) or int32 (30-bit mode), not {a.dtype}")


def _call(name, a, *args):
    """lfo_<name> / lfo30_<name> by the dtype of `a`; numpy arrays among args must have the same word dtype (int32 index
    tables are passed through _i32 by the caller), the token "N" stands for the row length of `a` in the mode's C type."""
    pre, W, NT = _mode(a)
    conv = []
    for x in (a,) + args:
        if isinstance(x, np.ndarray):
            assert x.dtype == W, (name, x.dtype, W)
            conv.append(_p(x))
        else:
            conv.append(x)
    return getattr(lib(), pre + name)(*conv)


def _N(a):
    return _mode(a)[2](a.shape[-1])


def mm_scalar(a, b, ql, qh, kl, kh, bits=62) -> int:
    f = lib().lfo_mm_scalar if bits == 62 else lib().lfo30_mm_scalar
    return int(f(a, b, ql, qh, kl, kh))


def redc_scalar(x, ql, qh, kl, kh, bits=62) -> int:
    f = lib().lfo_redc_scalar if bits == 62 else lib().lfo30_redc_scalar
    return int(f(x, ql, qh, kl, kh))


def mont_mult(a, b, c, rows, ql, qh, kl, kh):
    _call("mont_mult", a, b, c, int(rows), _N(a), ql, qh, kl, kh)


def mont_enter(a, Rs, rows, ql, qh, kl, kh):
    _call("mont_enter", a, Rs, int(rows), _N(a), ql, qh, kl, kh)


def mont_redc(a, rows, ql, qh, kl, kh):
    _call("mont_redc", a, int(rows), _N(a), ql, qh, kl, kh)


def ntt_tab(a, even, odd, psi, rows, _2q, ql, qh, kl, kh):
    logN = even.shape[0]
    _call("ntt_tab", a, _i32(even), _i32(odd), psi, int(rows), int(logN), _N(a), _2q, ql, qh, kl, kh)


def enter_ntt_tab(a, Rs, even, odd, psi, rows, _2q, ql, qh, kl, kh):
    logN = even.shape[0]
    _call("enter_ntt_tab", a, Rs, _i32(even), _i32(odd), psi, int(rows), int(logN), _N(a), _2q, ql, qh, kl, kh)


def intt_tab(a, even, odd, psi, Ninv, rows, _2q, ql, qh, kl, kh):
    logN = even.shape[0]
    _call("intt_tab", a, _i32(even), _i32(odd), psi, Ninv, int(rows), int(logN), _N(a), _2q, ql, qh, kl, kh)


def ntt(a, psi_br, rows, logN, _2q, ql, qh, kl, kh):
    assert a.shape[-1] == (1 << logN) and psi_br.shape[-1] == (1 << logN)
    _call("ntt", a, psi_br, int(rows), int(logN), _2q, ql, qh, kl, kh)


def intt(a, ipsi_br, Ninv, rows, logN, _2q, ql, qh, kl, kh):
    assert a.shape[-1] == (1 << logN) and ipsi_br.shape[-1] == (1 << logN)
    _call("intt", a, ipsi_br, Ninv, int(rows), int(logN), _2q, ql, qh, kl, kh)


def reduce_2q(a, rows, _2q):
    _call("reduce_2q", a, int(rows), _N(a), _2q)


def make_signed(a, rows, _2q):
    _call("make_signed", a, int(rows), _N(a), _2q)


def make_unsigned(a, rows, _2q):
    _call("make_unsigned", a, int(rows), _N(a), _2q)


def tile_unsigned(a, dst, rows, _2q):
    _call("tile_unsigned", a, dst, int(rows), _N(dst), _2q)


def mont_add(a, b, c, rows, _2q):
    _call("mont_add", a, b, c, int(rows), _N(a), _2q)


def mont_sub(a, b, c, rows, _2q):
    _call("mont_sub", a, b, c, int(rows), _N(a), _2q)


def place_rows(src, rows):
    """[rows, N] copy of src (rows repeat cyclically), every row first touched by the OpenMP thread that ntt / intt run it
    on (bench.py's cpu_baseline: page placement on a multi-socket host)."""
    assert src.dtype == np.int64 and src.ndim == 2
    dst = np.empty((rows, src.shape[1]), dtype=np.int64)
    lib().lfo_place_rows(_i64(dst), _i64(np.ascontiguousarray(src)), int(rows), int(src.shape[0]), _N(src))
    return dst


def omp_threads(n=None):
    """Read (n is None) or set the OpenMP thread count of the oracle's parallel loops."""
    gomp = ctypes.CDLL("libgomp.so.1")
    before = gomp.omp_get_max_threads()
    if n is not None:
        gomp.omp_set_num_threads(int(n))
    return before


def pin_threads(cpus):
    """Pin OpenMP thread t of a team of len(cpus) to logical CPU cpus[t] (the calling thread is thread 0: restore its
    affinity afterwards with os.sched_setaffinity).  Returns how many threads could not be pinned."""
    arr = (ctypes.c_int * len(cpus))(*cpus)
    return int(lib().lfo_pin_threads(arr, len(cpus)))


def galois(a, dst, rows, p):
    _call("galois", a, dst, int(rows), _N(a), _mode(a)[2](p))
