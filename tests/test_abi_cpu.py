"""The C-ABI library loads without a GPU and exports every symbol include/ckks_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "ckks_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint(?:64_t)?\s+(lf(?:30)?_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    from liberate_fhe_amd import _native
    names = declared()
    assert len(names) >= 19
    so = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/ckks_hip.h but not exported"
    assert sorted(_native.EXPORTED) == names, "python binding table out of sync with the header"
    assert _native.lib.lf_abi_version() == 15     # pure host call, no HIP runtime use


def test_python_binding_passes_as_many_arguments_as_the_header_declares():
    """ctypes does not check arity against the library: a signature that drifts from include/ckks_hip.h would shift every
    following argument.  Parameter counts (and pointer / integer kind) of the binding table against the header."""
    from liberate_fhe_amd import _native
    text = open(os.path.join(ROOT, "include", "ckks_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, params in re.findall(r"\bint(?:64_t)?\s+(lf(?:30)?_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = [p.strip() for p in params.split(",")] if params.strip() not in ("", "void") else []
        sig = _native._SIGNATURES[name]
        assert len(sig) == len(params), f"{name}: header has {len(params)} parameters, binding {len(sig)}"
        for i, (decl, ct) in enumerate(zip(params, sig)):
            is_ptr = "*" in decl
            bound_ptr = ct is ctypes.c_void_p or (isinstance(ct, type) and issubclass(ct, ctypes._Pointer))
            assert is_ptr == bound_ptr, f"{name} parameter {i} ({decl}): pointer-ness differs from the binding ({ct})"


def test_shim_exposes_the_fifteen_reference_functions():
    from liberate_fhe_amd.ntt import ntt_cuda
    expected = {"mont_mult", "mont_enter", "ntt", "enter_ntt", "intt", "mont_redc", "intt_exit", "intt_exit_reduce",
                "intt_exit_reduce_signed", "reduce_2q", "make_signed", "make_unsigned", "mont_add", "mont_sub",
                "tile_unsigned"}   # reference: src/liberate/ntt/ntt.cpp:421-437
    assert expected == set(ntt_cuda.__all__)
    for n in expected:
        assert callable(getattr(ntt_cuda, n))


def test_product_path_refuses_cpu_tensors():
    """No silent CPU fallback: the shim raises on host tensors."""
    import pytest
    import torch
    from liberate_fhe_amd.ntt import ntt_cuda
    t = torch.zeros((1, 8), dtype=torch.int64)
    with pytest.raises(RuntimeError):
        ntt_cuda.reduce_2q([t], [torch.zeros(1, dtype=torch.int64)])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "liberate_fhe_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("checker", ""), f


def test_every_include_and_every_source_file_is_part_of_the_library_digest():
    """build() rebuilds (and bench.py drops stamped profiles) when library_digest() changes: every file the library is
    compiled from must be in it — each #include "..." of csrc/ and each .hip / .h file that exists there."""
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, "liberate_fhe_amd", "csrc")
    known = {os.path.realpath(p) for p in g.library_sources()}
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h")):
            continue
        path = os.path.join(csrc, f)
        assert os.path.realpath(path) in known, f"{f} exists in csrc/ but is not a library source"
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
            target = os.path.realpath(os.path.join(csrc, inc))
            assert os.path.exists(target), f"{f} includes {inc}: no such file"
            assert target in known, f"{f} includes {inc}, which library_sources() does not list"


def test_library_reads_no_environment_variables():
    """A drop-in library's results must not depend on ambient env vars: no getenv anywhere in the C sources."""
    csrc = os.path.join(ROOT, "liberate_fhe_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_limits_are_exported_and_checked_by_the_engine():
    """lf_limits reports the capacities compiled into the fused kernels; the engine refuses a parameter set
    beyond them at construction (10 special primes: digits of 10 limbs > 8) instead of failing inside a launch."""
    import pytest
    from liberate_fhe_amd import _native
    from liberate_fhe_amd.fhe.backend import HipBackend
    from liberate_fhe_amd.fhe.presets import errors
    assert [_native.lib.lf_limits(i) for i in range(6)] == [8, 8, 250, 8, 24, -1]
    assert HipBackend.limits["special_primes"] == 8

    from liberate_fhe_amd.fhe import ckks_engine
    from tests.oracle_backend import OracleBackend
    ob = OracleBackend()
    ob.limits = HipBackend.limits
    with pytest.raises(errors.KernelLimitExceeded):
        ckks_engine(devices=["cpu"], backend=ob, logN=14, num_special_primes=10, num_scales=10, is_secured=False)


def test_unpickler_refuses_foreign_globals(tmp_path):
    """load() resolves only the container class and the tensor / array reconstructors."""
    import io
    import pickle
    import pytest
    from liberate_fhe_amd.fhe.evaluator import _PortableUnpickler

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        _PortableUnpickler(io.BytesIO(pickle.dumps(Evil()))).load()
    golden = os.path.join(ROOT, "tests", "golden", "reference_saved_ct.pkl")
    ct = _PortableUnpickler(open(golden, "rb")).load()   # a file the reference wrote still loads
    assert ct.origin == "cipher text"


def test_relaxed_transforms_need_table_and_host_primes():
    """LF_NTT_RELAXED with psi_dp == NULL or q_host == NULL is LF_ERR_ARG — refused from the arguments
    alone, before any device call (so it can be checked without a GPU) — and the Python Consts refuses the same."""
    import ctypes
    import numpy as np
    import pytest
    import torch
    from liberate_fhe_amd._native import lib
    from liberate_fhe_amd.fhe.backend import Consts
    dummy = ctypes.c_void_p(64)          # never dereferenced: every call below fails its argument check
    q = np.array([1099511922689], dtype=np.int64)
    for psi_dp, q_host in ((None, q.ctypes.data), (dummy, None), (None, None)):
        assert lib.lf_ntt(dummy, 1, 1, 13, dummy, psi_dp, q_host, None, 1, dummy, dummy, dummy, dummy, dummy, 0, None) == 10001
        assert lib.lf_intt(dummy, 1, 1, 13, dummy, psi_dp, q_host, dummy, 2, 1, dummy, dummy, dummy, dummy, dummy, 0, None) == 10001
        arr = (ctypes.c_void_p * 1)(64)
        assert lib.lf_rescale_ntt(arr, arr, 1, dummy, 1, 13, dummy, 0, dummy, psi_dp, q_host, None, 1, dummy, dummy, dummy,
                                  dummy, dummy, 0, None) == 10001
    assert lib.lf_intt(dummy, 1, 1, 13, dummy, dummy, q.ctypes.data, dummy, 1, 1, dummy, dummy, dummy, dummy, dummy, 0, None) == 10001
    z = torch.zeros(1, dtype=torch.int64)
    c = Consts.__new__(Consts)
    c._qptr = 0
    with pytest.raises(ValueError):
        c.qptr(relaxed=True)


def test_unpickler_accepts_what_the_reference_load_accepts(tmp_path):
    """Python complex, ndarrays pickled at protocols 2 and 5 inside a container still load."""
    import io
    import pickle
    import numpy as np
    import torch
    from liberate_fhe_amd.fhe.data_struct import data_struct
    from liberate_fhe_amd.fhe.evaluator import _PortableUnpickler
    payload = data_struct(data=[np.arange(6, dtype=np.int64).reshape(2, 3), 1.5 + 2j, torch.arange(4)], include_special=False,
                          ntt_state=False, montgomery_state=False, origin="cipher text", level=0, hash="h", version="v")
    for proto in (2, 4, 5):
        back = _PortableUnpickler(io.BytesIO(pickle.dumps(payload, protocol=proto))).load()
        assert (back.data[0] == payload.data[0]).all() and back.data[1] == 1.5 + 2j and torch.equal(back.data[2], payload.data[2])


def test_lf_tune_is_a_pure_host_call():
    """Launch-shape thresholds: read, set, restore, unknown knob -> -1; no device is touched."""
    from liberate_fhe_amd._native import lib
    assert lib.lf_tune(0, -1) == -1     # the round-3 one-launch knob is gone
    cols = lib.lf_tune(1, -1)
    assert 0 <= cols <= 5
    assert lib.lf_tune(1, 9) == cols and lib.lf_tune(1, -1) == cols      # out of range: ignored
    assert lib.lf_tune(3, -1) == 1 and lib.lf_tune(3, 5) == 1     # digit planes: on by default; out of range: ignored
    assert lib.lf_tune(4, -1) == 1 and lib.lf_tune(4, 7) == 1     # extra column stage of lf_ntt_ws: on; out of range: ignored
    assert lib.lf_tune(5, -1) == 3 and lib.lf_tune(5, 9) == 3     # planes for the sums (bit 0) and the operand stack (bit 1): on
    assert lib.lf_tune(77, 1) == -1


def test_stack_planes_query_follows_the_primes_and_the_knobs():
    """lf_stack_planes (pure host): internal stacks keep fp64-class rows as 6-byte planes at two-pass ring degrees when the rows
    hold primes of BOTH classes and lf_tune's LF_TUNE_DIGIT_PLANES / LF_TUNE_MORE_PLANES (bit 1) say so."""
    import numpy as np
    from liberate_fhe_amd._native import lib
    small, large = (1 << 40) + 12345, (1 << 60) - 93
    mixed = np.array([small, small + 2, large], dtype=np.int64)
    only_small = np.array([small, small + 2], dtype=np.int64)
    only_large = np.array([large], dtype=np.int64)
    q = lambda a: a.ctypes.data
    assert lib.lf_stack_planes(16, 3, q(mixed)) == 1 and lib.lf_stack_planes(13, 3, q(mixed)) == 1
    assert lib.lf_stack_planes(12, 3, q(mixed)) == 0                      # one-pass ring degree: no stack crosses HBM twice
    assert lib.lf_stack_planes(16, 2, q(mixed)) == 0                      # the first two rows only: one class
    assert lib.lf_stack_planes(16, 2, q(only_small)) == 0 and lib.lf_stack_planes(16, 1, q(only_large)) == 0
    assert lib.lf_stack_planes(16, 3, None) == 0
    try:
        lib.lf_tune(5, 1)
        assert lib.lf_stack_planes(16, 3, q(mixed)) == 0
        lib.lf_tune(5, 3)
        lib.lf_tune(3, 0)
        assert lib.lf_stack_planes(16, 3, q(mixed)) == 0
    finally:
        lib.lf_tune(3, 1)
        lib.lf_tune(5, 3)
    assert lib.lf_stack_planes(16, 3, q(mixed)) == 1


def test_clock_probe_checks_its_arguments_on_the_host():
    """lf_clock_probe (measurement entry): bad arguments are refused before any HIP call."""
    from liberate_fhe_amd._native import lib
    assert lib.lf_clock_probe(None, 4, 1000, 0, None) != 0
    assert lib.lf_clock_probe(8, 0, 1000, 0, None) != 0
    assert lib.lf_clock_probe(8, 4, 0, 0, None) != 0
    assert lib.lf_clock_probe(8, 4096, 1_000_000, 0, None) != 0           # more than ten seconds of spinning


def test_bench_refuses_more_ranks_than_gpus_without_touching_one():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks — after counting the GPUs WITHOUT
    initialising one; on a box with fewer than N it says so and exits 2 (no rank is started, nothing hangs)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LF_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s)" in r.stderr and r.stdout.strip() == ""


def test_hot_kernels_use_no_scratch_memory_and_the_tracked_table_is_current():
    """The design argues from register counts and occupancies: they come from the compiler's own
    kernel-resource-usage remarks of the objects the library is linked from (__graft_entry__.kernel_resources()) and are tracked as
    profiles/r06_kernel_resources.txt.  Held here: no stack object and no spill (scratch = 0 bytes per lane) in the kernels
    every preset's hot path launches — the column passes, the 16-words-per-thread tiled passes, the key switch's extension,
    inner product and digit kernels (a 16-byte struct selected between registers and global memory once put 48 .. 128 bytes per
    lane of ks_inner2_kernel<2 | 4, fold> into scratch; a rolled loop 272 bytes of ntt_inv_cols_digits) — and the occupancies the
    launch shapes were chosen for."""
    import re
    import __graft_entry__ as g
    rows = g.kernel_resources()
    assert len(rows) > 100
    by = {}
    for r in rows:
        by.setdefault(r["kernel"], []).append(r)
    hot = re.compile(r"^(ntt_fwd_cols_ws<|ntt_fwd_cols_mixed|ntt_inv_cols_mixed<|ntt_inv_cols_ws<|ntt_inv_cols_digits<|ntt_pass16|"
                     r"ks_ext_cols_mixed<|ks_inner2_kernel<|ks_digits_kernel|ks_moddown|ks_pivots|ew_kernel<|galois_kernel)")
    seen = [k for k in by if hot.match(k)]
    for need in ("ntt_fwd_cols_ws<5>", "ntt_pass16_fwd_seq_ws<true>", "ntt_pass16_fwd_seq_ws<false>", "ntt_pass16_fwd_planes",
                 "ks_ext_cols_mixed<4>", "ks_inner2_kernel<1, true, true, true>", "ks_inner2_kernel<4, true, true, true>"):
        assert need in by, (need, sorted(seen)[:20])
    bad = {k: [(r["file"], r["scratch"], r["vgpr_spill"]) for r in by[k]] for k in seen if any(r["scratch"] or r["vgpr_spill"] for r in by[k])}   # (SGPR spills go to VGPR lanes, not to memory)
    assert not bad, bad
    # what the launch shapes rest on: 4 tiles of 34 KiB LDS per CU for the tiled passes, 3 waves per SIMD for the five-stage column pass
    for k in seen:
        if k.startswith("ntt_pass16"):
            assert all(r["occupancy"] >= 4 and r["lds"] <= 40960 for r in by[k]), (k, by[k])
    assert all(r["occupancy"] >= 3 for r in by["ntt_fwd_cols_ws<5>"])
    assert all(r["occupancy"] >= 4 for r in by["ks_inner2_kernel<4, true, true, true>"])
    # the only kernels with scratch at all are the 8-words tiled passes (logN <= 12 / in-place logN 17), capped at 80 VGPRs for 6 waves
    allowed = re.compile(r"^(ntt_fwd_pass<|ntt_inv_pass_io<|ntt_inv_pass_mixed<)")
    other = sorted({r["kernel"] for r in rows if r["scratch"] and not allowed.match(r["kernel"])})
    assert not other, other
    assert max(r["scratch"] for r in rows) <= 64
    # the tracked table is the current build's (tools/kernel_resources.py writes it)
    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.txt")
    tracked = {}
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("file "):
            continue
        f = ln.split()
        tracked.setdefault(" ".join(f[1:-7]), set()).add((int(f[-7]), int(f[-3]), int(f[-1])))      # VGPR, scratch, occupancy
    stale = {k: (sorted(tracked.get(k, [])), sorted({(r["vgprs"], r["scratch"], r["occupancy"]) for r in by[k]})) for k in seen
             if len(k) <= 57 and tracked.get(k) != {(r["vgprs"], r["scratch"], r["occupancy"]) for r in by[k]}}
    assert not stale, f"profiles/r06_kernel_resources.txt is stale (python tools/kernel_resources.py r06): {stale}"
