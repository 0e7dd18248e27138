"""ctypes binding of libckks_hip.so (C ABI declared in include/ckks_hip.h).

There is NO fallback: if the HIP library has not been built (python -c "import __graft_entry__ as g;
g.build()"), importing this module raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LF_HIP_LIB") or os.path.join(_HERE, "csrc", "libckks_hip.so")  # env override: kernel experiments

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. "
        "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path."
    )

lib = ctypes.CDLL(LIB_PATH)

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_U = ctypes.c_uint64

_SIGNATURES = {
    "lf_abi_version": [],
    "lf_limits": [_I],
    "lf_tune": [_I, _I],
    "lf_stack_planes": [_I, _I, _P],
    "lf_clock_probe": [_P, _I, _U, _I, _P],
    "lf_ntt_pass": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "lf_ntt_ws_words": [_I, _I, _I],
    "lf_ntt_ws": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "lf_intt_ws": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "lf_ntt_pass_ws": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "lf_mont_mult": [_P, _P, _P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf_mont_enter": [_P, _P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf_mont_redc": [_P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf_reduce_2q": [_P, _I, _L, _P, _I, _P],
    "lf_make_signed": [_P, _I, _L, _P, _I, _P],
    "lf_make_unsigned": [_P, _I, _L, _P, _I, _P],
    "lf_tile_unsigned": [_P, _P, _I, _L, _P, _I, _P],
    "lf_mont_add": [_P, _P, _P, _I, _L, _P, _I, _P],
    "lf_mont_sub": [_P, _P, _P, _I, _L, _P, _I, _P],
    "lf_twiddle_dp": [_P, _P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf_ntt": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P],
    "lf_intt": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "lf_galois": [_P, _P, _I, _I, _L, _P, _I, _P],
    "lf_rescale": [_P, _P, _P, _I, _L, _P, _L, _P, _P, _P, _P, _I, _P],
    "lf_tensor": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _I, _P],
    "lf_ks_digits": [_P, _P, _I, _P, _P, _L, _P, _P, _P, _P, _I, _P],
    "lf_ks_extend": [_P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_inner": [_P, _P, _L, _L, _L, _P, _P, _I, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf_ks_moddown": [_P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_rescale_batch": [_P, _P, _P, _I, _I, _L, _P, _L, _P, _P, _P, _P, _I, _P],
    "lf_ks_moddown_batch": [_P, _P, _P, _I, _I, _I, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_moddown_ws": [_P, _P, _P, _I, _I, _I, _L, _P, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_moddown_ws_words": [_I, _I, _I, _L],
    "lf_ks_moddown_consts": [_P, _L, _I, _I, _I, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_moddown_one": [_P, _P, _P, _I, _I, _I, _L, _P, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_digits_batch": [_P, _P, _I, _I, _P, _P, _L, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_digits_galois": [_P, _P, _I, _P, _P, _L, _L, _P, _P, _P, _P, _P, _I, _P],
    "lf_galois_batch": [_P, _P, _I, _I, _I, _L, _P, _I, _P],
    "lf_gather_rows": [_P, _P, _I, _L, _I, _P],
    "lf_rescale_ntt": [_P, _P, _I, _P, _I, _I, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P],
    "lf_chacha20": [_P, _P, _L, _U, _I, _P],
    "lf_randint_fast": [_P, _P, _I, _L, _P, _L, _U, _I, _P],
    "lf_randint": [_P, _I, _L, _P, _I, _P],
    "lf_discrete_gaussian_fast": [_P, _P, _L, _P, _I, _I, _U, _I, _P],
    "lf_discrete_gaussian": [_P, _L, _P, _I, _I, _I, _P],
    "lf_randround": [_P, _P, _L, _I, _P],
    "lf_key_planes": [_P, _P, _P, _P, _I, _L, _P, _P, _I, _P],
    "lf_ks_core_batch": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_fwd": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_tail": [_I, _I, _I, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_intt_mul": [_P, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "lf_intt_mul_digits": [_P, _P, _L, _P, _L, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "lf_relin_core_batch": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_relin_fwd": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_relin_tail": [_I, _I, _I, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf_ks_core": [_P, _I, _I, _I, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
}



class KsPlan(ctypes.Structure):
    """lf_ks_plan (include/ckks_hip.h), field for field."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("logN", "ell", "K", "nparts", "dig_nparts", "device", "max_nct", "md_consts")]
                + [("round_at", _L), ("md_ws_words", _L)]
                + [(n, _P) for n in ("ql", "qh", "kl", "kh", "_2q", "Rs", "Ninv", "q_host", "psi", "ipsi", "psi_dp", "ipsi_dp",
                                     "dig_desc", "dig_tab", "ext_desc", "E", "Ed", "PiR", "PiP", "own", "rescale_scales", "PR",
                                     "state", "ext", "sum", "md_ws", "x4", "d2")])


_SIGNATURES["lf_cc_mult_evk"] = [ctypes.POINTER(KsPlan), _P, _P, _P, _L, _L, _L, _I, _P, _P, _P]
_SIGNATURES["lf_switch_key"] = [ctypes.POINTER(KsPlan), _P, _P, _L, _I, _P, _L, _L, _L, _I, _P, _P, _P]

_PL = ctypes.POINTER(KsPlan)
_SIGNATURES["lf_switch_key_batch"] = [_PL, _I, _P, _P, _L, _I, _P, _L, _L, _L, _I, _P, _P, _P]
_SIGNATURES["lf_cc_mult_evk_batch"] = [_PL, _I, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P]
_SIGNATURES["lf_cc_mult_evk_pre"] = [_PL, _P, _P, _I, _P]
_SIGNATURES["lf_switch_key_pre"] = [_PL, _P, _L, _I, _P]
_SIGNATURES["lf_ks_plan_fwd"] = [_PL, _P, _I, _I, _I, _P]
_SIGNATURES["lf_cc_mult_evk_post"] = [_PL, _P, _L, _L, _L, _I, _P, _P, _I, _P]
_SIGNATURES["lf_switch_key_post"] = [_PL, _P, _L, _I, _P, _L, _L, _L, _I, _P, _P, _I, _P]

# the 30-bit / int32 word mode of the ntt_cuda surface (csrc/ckks_w30.hip)
_SIGNATURES.update({
    "lf30_mont_mult": [_P, _P, _P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf30_mont_enter": [_P, _P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf30_mont_redc": [_P, _I, _L, _P, _P, _P, _P, _I, _P],
    "lf30_reduce_2q": [_P, _I, _L, _P, _I, _P],
    "lf30_make_signed": [_P, _I, _L, _P, _I, _P],
    "lf30_make_unsigned": [_P, _I, _L, _P, _I, _P],
    "lf30_tile_unsigned": [_P, _P, _I, _L, _P, _I, _P],
    "lf30_mont_add": [_P, _P, _P, _I, _L, _P, _I, _P],
    "lf30_mont_sub": [_P, _P, _P, _I, _L, _P, _I, _P],
    "lf30_ntt": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lf30_intt": [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P],
})

for _name, _args in _SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int64 if _name.endswith("_words") else ctypes.c_int


class HipError(RuntimeError):
    pass


def check(code: int, what: str):
    if code == LF_ERR_STATE:
        raise HipError(f"{what} refused (status {code}, LF_ERR_STATE): the scratch the first half of this operation left is in another "
                       "format than this half reads — lf_tune changed between the two; repeat the first half")
    if code != 0:
        raise HipError(f"{what} failed with status {code}")


EXPORTED = tuple(_SIGNATURES)
LF_KEY_RAW = 0
LF_KEY_PLANES = 1
LF_NTT_RELAXED = 1
LF_NTT_PLAIN = 2
LF_NTT_ONLY_COLS = 4
LF_NTT_ONLY_TILED = 8
LF_NTT_PLANES = 16
LF_STACK_PLANES = 4
LF_ERR_ARG = 10001
LF_ERR_STATE = 10002
