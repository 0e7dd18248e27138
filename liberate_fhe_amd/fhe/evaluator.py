"""The evaluator surface around the hot path: data movement and (de)serialisation, scalar and
plaintext operands, negation, slot statistics, and the multiparty protocol helpers — mixed into
`ckks_engine` (reference: src/liberate/fhe/ckks_engine.py, `eng.py` below).

Every method here is a composition of the `ntt_context` wrappers and of the engine's hot methods
(rescale / cc_mult / relinearize / rotate_single), issued in the reference's order with the
reference's constants, so results are word-for-word the reference's; none has a kernel of its own.
"""
from __future__ import annotations

import datetime
import io
import math
import pickle
import sys
import types as pytypes
from pathlib import Path

import numpy as np
import torch

from . import encdec
from .data_struct import data_struct
from .presets import errors, types

_REFERENCE_DS_MODULE = "liberate.fhe.data_struct"


def _safe_storage_from_bytes(blob):
    """What torch.storage._load_from_bytes does, minus the arbitrary-code path: the nested stream is read by
    torch's restricted (weights_only) unpickler."""
    return torch.load(io.BytesIO(blob), weights_only=True)


# Every global a container file can legitimately name.  Files are exchanged between parties (client keys,
# server results), so an untrusted .pkl is the NORMAL input: anything outside this list is refused instead of
# being imported and called, which is what pickle.load would do (the reference's load() has that hole, eng.py:2024-2029).
def _np_core():
    """numpy's private core package: `numpy._core` from 1.26.1 / 2.x on, `numpy.core` before."""
    return getattr(np, "_core", None) or np.core


_SAFE_GLOBALS = {
    ("torch._utils", "_rebuild_tensor_v2"): lambda: torch._utils._rebuild_tensor_v2,
    ("torch.storage", "_load_from_bytes"): lambda: _safe_storage_from_bytes,
    ("collections", "OrderedDict"): lambda: __import__("collections").OrderedDict,
    ("builtins", "complex"): lambda: complex,                       # a Python complex inside a container
    ("__builtin__", "complex"): lambda: complex,                    # .. as pickle protocols <= 2 spell the module
    ("_codecs", "encode"): lambda: __import__("_codecs").encode,    # ndarray payloads of pickle protocols <= 2
    ("numpy", "dtype"): lambda: np.dtype,
    ("numpy", "ndarray"): lambda: np.ndarray,
}
# numpy's reconstructors under both spellings of the core package (files written by numpy 1.x and 2.x); protocol 5
# writes ndarrays through numeric._frombuffer
for _mod in ("numpy._core", "numpy.core"):
    _SAFE_GLOBALS[(_mod + ".multiarray", "scalar")] = lambda: _np_core().multiarray.scalar
    _SAFE_GLOBALS[(_mod + ".multiarray", "_reconstruct")] = lambda: _np_core().multiarray._reconstruct
    _SAFE_GLOBALS[(_mod + ".numeric", "_frombuffer")] = lambda: _np_core().numeric._frombuffer
for _name in ("LongStorage", "DoubleStorage", "FloatStorage", "IntStorage", "ShortStorage", "CharStorage", "ByteStorage",
              "BoolStorage", "HalfStorage", "ComplexDoubleStorage", "ComplexFloatStorage", "UntypedStorage"):
    if hasattr(torch, _name):
        _SAFE_GLOBALS[("torch", _name)] = (lambda n=_name: getattr(torch, n))


class _PortableUnpickler(pickle.Unpickler):
    """Reads containers pickled by this package or by the reference (whose class lives in
    liberate.fhe.data_struct): both are field-for-field the same NamedTuple.  Only the globals such a file needs
    (the container class, torch tensor / storage reconstructors, numpy array / scalar reconstructors) resolve;
    any other raises pickle.UnpicklingError."""

    def find_class(self, module, name):
        if name == "data_struct" and module in (_REFERENCE_DS_MODULE, data_struct.__module__):
            return data_struct
        hit = _SAFE_GLOBALS.get((module, name))
        if hit is None:
            raise pickle.UnpicklingError(f"refusing to load global {module}.{name} from a ciphertext / key file")
        return hit()


def _portable_dumps(obj) -> bytes:
    """Pickle with data_structs recorded under the reference's class path, so files written here load in
    the reference (eng.py:2017-2029) and the other way round."""
    def _make(*fields):
        return data_struct(*fields)
    _make.__module__, _make.__qualname__, _make.__name__ = _REFERENCE_DS_MODULE, "data_struct", "data_struct"

    class _Pickler(pickle.Pickler):
        def reducer_override(self, o):
            if is_struct(o):
                return _make, tuple(o)
            return NotImplemented

    # pickle checks that the recorded path resolves to the object it is writing; lend it one for the call
    chain = _REFERENCE_DS_MODULE.split(".")
    names = [".".join(chain[:i + 1]) for i in range(len(chain))]
    saved = {n: sys.modules.get(n) for n in names}
    try:
        for n in names:
            sys.modules[n] = pytypes.ModuleType(n)
        sys.modules[_REFERENCE_DS_MODULE].data_struct = _make
        buf = io.BytesIO()
        _Pickler(buf, protocol=pickle.DEFAULT_PROTOCOL).dump(obj)
        return buf.getvalue()
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def is_struct(x):
    """A data_struct of this package, or the reference's field-for-field identical NamedTuple."""
    return isinstance(x, data_struct) or getattr(x, "_fields", None) == data_struct._fields


class EvaluatorOps:
    # =============================================================================================
    # clone / host <-> device movement (eng.py:1740-1927)
    # =============================================================================================
    def clone_tensors(self, data):
        if not isinstance(data[0], list):
            return [t.clone() for t in data]
        return [[t.clone() for t in part] for part in data]

    def clone(self, text):
        """Deep copy; tensor containers come back as lists (the reference's clone turns tuples into lists,
        and mc_mult / mc_add rely on that to assign components)."""
        def rec(x):
            if isinstance(x, torch.Tensor):
                return x.clone()
            if hasattr(x, "_replace") and hasattr(x, "data"):   # data_struct (this package's or a foreign one)
                return x._replace(data=rec(x.data))
            if isinstance(x, (list, tuple)):
                return [rec(y) for y in x]
            return x
        return rec(text)

    def _dest_rows(self, level, include_special):
        dest = (self.ntt.p.destination_arrays_with_special if include_special else self.ntt.p.destination_arrays)[level]
        lo = min(min(d) for d in dest if len(d))
        return [[i - lo for i in d] for d in dest]

    def download_to_cpu(self, gpu_data, level, include_special):
        """Per-device row blocks -> ONE host tensor with the rows in natural prime order (eng.py:1790-1821).
        With one process per GPU the blocks of the other ranks arrive through an all-gather."""
        dest = self._dest_rows(level, include_special)
        alive = [d for d in range(len(dest)) if len(dest[d])]
        loc = [d for d in alive if d in self.local_ids]
        for t in gpu_data:
            if t.device.type != "cuda" and not getattr(self.backend, "host_tensors", False):
                raise Exception("To download data to the CPU, it must already be in a GPU!!!")
        # The row count follows the reference (sum over GPUs), which counts the replicated special rows once
        # per GPU: with several GPUs the host tensor carries that many unused trailing rows (zeros here,
        # uninitialised memory in the reference); upload_to_gpu never reads them.
        out = torch.zeros((sum(len(d) for d in dest), self.ctx.N), dtype=torch.int64, device="cpu")
        if getattr(self, "_multi", False):
            max_rows = max(len(dest[d]) for d in alive)
            mine = torch.zeros((max_rows, self.ctx.N), dtype=torch.int64, device=self.ntt.devices[self.local_ids[0]])
            if loc:
                mine[:gpu_data[0].size(0)] = gpu_data[0]
            for d, block in enumerate(self.comm.all_gather(mine)):
                if d in alive:
                    out[dest[d]] = block[:len(dest[d])].cpu()
        else:
            for t, d in zip(gpu_data, loc):
                out[dest[d]] = t.cpu()
        return [out]

    def upload_to_gpu(self, cpu_data, level, include_special):
        host = cpu_data[0]
        if host.device.type != "cpu":
            raise Exception("To upload data to GPUs, it must already be in the CPU!!!")
        dest = self._dest_rows(level, include_special)
        return [host[dest[d]].to(device=self.ntt.devices[d]) for d in range(len(dest)) if d in self.local_ids and len(dest[d])]

    def move_tensors(self, data, level, include_special, direction):
        func = {"gpu2cpu": self.download_to_cpu, "cpu2gpu": self.upload_to_gpu}[direction]
        if not isinstance(data[0], (list, tuple)):
            return func(data, level, include_special)
        return [func(part, level, include_special) for part in data]

    def move_to(self, text, direction="gpu2cpu"):
        if not is_struct(text.data[0]):
            return text._replace(data=self.move_tensors(text.data, text.level, text.include_special, direction))
        return text._replace(data=[self.move_to(d, direction) for d in text.data])

    def cpu(self, ct):
        return self.move_to(ct, "gpu2cpu")

    def cuda(self, ct):
        return self.move_to(ct, "cpu2gpu")

    def tensor_device(self, data):
        return (data[0] if not isinstance(data[0], (list, tuple)) else data[0][0]).device.type

    def device(self, text):
        if not is_struct(text.data[0]):
            return self.tensor_device(text.data)
        return self.device(text.data[0])

    # =============================================================================================
    # printing (eng.py:1930-1992)
    # =============================================================================================
    def tree_lead_text(self, level, tabs=2, final=False):
        if level == 0:
            return "─" * tabs + "┬" + "─" * tabs
        if level < 0:
            return " " * tabs + "│" * (-level - 1) + ("└" if final else "├") + "─" * tabs
        return " " * tabs + "│" * (level - 1) + "├" + "┬" + "─" * (tabs - 1)

    def print_data_shapes(self, data, level):
        parts = data if isinstance(data[0], (list, tuple)) else [data]
        for pi, part in enumerate(parts):
            for di, t in enumerate(part):
                last = pi == len(parts) - 1 and di == len(part) - 1
                print(f"{self.tree_lead_text(-level, final=last)} tensor at device {t.device} with shape {t.shape}.")

    def print_data_structure(self, text, level=0):
        print(f"{self.tree_lead_text(level)} {text.origin}")
        if not is_struct(text.data[0]):
            self.print_data_shapes(text.data, level + 1)
        else:
            for d in text.data:
                self.print_data_structure(d, level + 1)

    # =============================================================================================
    # save / load (eng.py:1998-2029)
    # =============================================================================================
    def auto_generate_filename(self, fmt_str="%Y%m%d%H%M%s%f"):
        return datetime.datetime.now().strftime(fmt_str) + ".pkl"

    def save(self, text, filename=None, on_host=None):
        """Pickle of the host form (one natural-order [rows, N] tensor per component), readable by the
        reference's load().  `on_host`: pass True if `text` is already the output of cpu()."""
        if filename is None:
            filename = self.auto_generate_filename()
        if on_host is None:
            on_host = self.device(text) == "cpu" and not getattr(self.backend, "host_tensors", False)
        host = text if on_host else self.cpu(text)   # collective on a multi-rank engine: every rank takes part
        comm = getattr(self, "comm", None)
        if comm is None or comm.world_size == 1 or comm.rank == 0:   # .. and exactly one of them writes the file
            Path(filename).write_bytes(_portable_dumps(host))

    def load(self, filename, move_to_gpu=True):
        with Path(filename).open("rb") as f:
            host = _PortableUnpickler(f).load()
        return self.cuda(host) if move_to_gpu else host

    # =============================================================================================
    # negate, scalar operands (eng.py:2035-2157)
    # =============================================================================================
    def negate(self, ct: data_struct) -> data_struct:
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        out = self.clone(ct)
        for part in out.data:
            for t in part:
                t *= -1
            self.ntt.make_signed(part, ct.level)
        return out

    def _row_scalars(self, value, level, montgomery):
        """value (a Python int) as one residue per local row, times R when `montgomery`."""
        v = value * self.ctx.R if montgomery else value
        return [self._t64([v % self.ctx.q[i] for i in self.ntt.p.destination_arrays[level][d]], d)
                for d in self._loc(level)]

    def _scale_rows(self, ct, scalars):
        out = self.clone(ct)
        for comp in (0, 1):
            self.ntt.mont_enter_scalar(out.data[comp], scalars, ct.level)
            self.ntt.reduce_2q(out.data[comp], ct.level)
        return out

    def mult_int_scalar(self, ct: data_struct, scalar, evk=None, relin=True):
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        return self._scale_rows(ct, self._row_scalars(int(scalar), ct.level, True))

    def mult_scalar(self, ct, scalar, evk=None, relin=True):
        scaled = int(scalar * self.scale * np.sqrt(self.deviations[ct.level + 1]) + 0.5)
        return self.rescale(self._scale_rows(ct, self._row_scalars(scaled, ct.level, True)))

    def add_scalar(self, ct, scalar):
        scaled = int(scalar * self.scale * self.deviations[ct.level] + 0.5)
        if self.norm == "backward":
            scaled *= self.ctx.N
        scaled *= self.int_scale
        out = self.clone(ct)
        for t, s in zip(out.data[0], self._row_scalars(scaled, ct.level, False)):
            t[:, 0] += s
        self.ntt.reduce_2q(out.data[0], ct.level)
        return out

    def sub_scalar(self, ct, scalar):
        return self.add_scalar(ct, -scalar)

    def int_scalar_mult(self, scalar, ct, evk=None, relin=True):
        return self.mult_int_scalar(ct, scalar)

    def scalar_mult(self, scalar, ct, evk=None, relin=True):
        return self.mult_scalar(ct, scalar)

    def scalar_add(self, scalar, ct):
        return self.add_scalar(ct, scalar)

    def scalar_sub(self, scalar, ct):
        return self.add_scalar(self.negate(ct), scalar)

    # =============================================================================================
    # plaintext (message) operands (eng.py:2165-2219)
    # =============================================================================================
    def mc_mult(self, m, ct, evk=None, relin=True):
        m = np.array(m) * np.sqrt(self.deviations[ct.level + 1])
        pt = self.ntt.tile_unsigned(self.encode(m, 0), ct.level)
        self.ntt.enter_ntt(pt, ct.level)
        out = self.clone(ct)
        self.ntt.enter_ntt(out.data[0], ct.level)
        self.ntt.enter_ntt(out.data[1], ct.level)
        d0 = self.ntt.mont_mult(pt, out.data[0], ct.level)
        d1 = self.ntt.mont_mult(pt, out.data[1], ct.level)
        self.ntt.intt_exit_reduce(d0, ct.level)
        self.ntt.intt_exit_reduce(d1, ct.level)
        return self.rescale(out._replace(data=[d0, d1]))

    def mc_add(self, m, ct):
        pt = self.ntt.tile_unsigned(self.encode(m, ct.level), ct.level)
        self.ntt.mont_enter_scale(pt, ct.level)
        out = self.clone(ct)
        self.ntt.mont_enter(out.data[0], ct.level)
        d0 = self.ntt.mont_add(pt, out.data[0], ct.level)
        self.ntt.mont_redc(d0, ct.level)
        self.ntt.reduce_2q(d0, ct.level)
        return out._replace(data=[d0, out.data[1]])

    def mc_sub(self, m, ct):
        return self.mc_add(m, self.negate(ct))

    def cm_mult(self, ct, m, evk=None, relin=True):
        return self.mc_mult(m, ct)

    def cm_add(self, ct, m):
        return self.mc_add(m, ct)

    def cm_sub(self, ct, m):
        return self.mc_add(-np.array(m), ct)

    # =============================================================================================
    # misc + slot statistics (eng.py:2289-2383, 2693-2724)
    # =============================================================================================
    def refresh(self):
        self.rng.refresh()

    def reduce_error(self, ct):
        return self.mult_scalar(ct, 1.0)

    def _fold_slots(self, ct, gk):
        for roti in range(self.ctx.logN - 1):
            ct = self.add(self.rotate_single(ct, gk.data[roti]), ct)
        return ct

    def sum(self, ct, gk):
        return self._fold_slots(self.clone(ct), gk)

    def mean(self, ct, gk, alpha=1):
        return self._fold_slots(self.mult(1 / self.num_slots / alpha, ct), gk)

    def cov(self, ct_a: data_struct, ct_b: data_struct, evk: data_struct, gk: data_struct) -> data_struct:
        dev_a = self.sub(ct_a, self.mean(ct_a, gk))
        dev_b = self.sub(ct_b, self.mean(ct_b, gk))
        return self.mult(self.mult(dev_a, dev_b, evk), 1 / (self.num_slots - 1))

    def pow(self, ct: data_struct, power: int, evk: data_struct) -> data_struct:
        """Square-and-multiply over the binary expansion of `power` (eng.py:2333-2351)."""
        squares, exponent = [ct], 2
        while exponent <= power:
            squares.append(self.cc_mult(squares[-1], squares[-1], evk))
            exponent *= 2
        out, remaining = squares[-1], power - exponent // 2
        while remaining > 0:
            ind = math.floor(math.log2(remaining))
            out = self.auto_cc_mult(out, squares[ind], evk)
            remaining -= 2 ** ind
        return out

    def sqrt(self, ct: data_struct, evk: data_struct, e=0.0001, alpha=0.0001) -> data_struct:
        """Wilkes-style coupled iteration on (a, b) with per-step cubic-root constants (eng.py:2693-2710)."""
        a, b = self.clone(ct), self.clone(ct)
        while e <= 1 - alpha:
            k = float(np.roots([1 - e ** 3, -6 + 6 * e ** 2, 9 - 9 * e])[1])
            b0 = self.sub_scalar(self.mult_scalar(a, k, evk), 3)
            b1 = self.mult_scalar(b, (k ** 0.5) / 2, evk)
            b = self.cc_mult(b0, b1, evk)
            a0 = self.mult_scalar(a, (k ** 3) / 4)
            a1 = self.square(self.sub_scalar(a, 3 / k), evk)
            a = self.cc_mult(a0, a1, evk)
            e = k * (3 - k) ** 2 / 4
        return b

    def var(self, ct: data_struct, evk: data_struct, gk: data_struct, relin=False) -> data_struct:
        dev = self.square(ct=self.sub(ct, self.mean(ct=ct, gk=gk)), evk=evk, relin=relin)
        if not relin:
            dev = self.relinearize(ct_triplet=dev, evk=evk)
        return self.mean(ct=dev, gk=gk)

    def std(self, ct: data_struct, evk: data_struct, gk: data_struct, relin=False) -> data_struct:
        return self.sqrt(ct=self.var(ct=ct, evk=evk, gk=gk, relin=relin), evk=evk)

    # =============================================================================================
    # multiparty (eng.py:2388-2690)
    # =============================================================================================
    def multiparty_public_crs(self, pk: data_struct):
        return self.clone(pk).data[1]

    def multiparty_create_public_key(self, sk: data_struct, a=None, include_special=False) -> data_struct:
        return self.create_public_key(sk, include_special=include_special, a=a)

    def multiparty_create_collective_public_key(self, pks: list[data_struct]) -> data_struct:
        first = pks[0]
        mult_type = -2 if first.include_special else -1
        b = [t.clone() for t in first.data[0]]
        a = [t.clone() for t in first.data[1]]
        for pk in pks[1:]:
            b = self.ntt.mont_add(b, pk.data[0], lvl=0, mult_type=mult_type)
        return first._replace(data=(b, a), origin=types.origins["pk"], hash=self.hash, version=self.version)

    def _check_partial_decrypt(self, ct, sk):
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        if ct.ntt_state or ct.montgomery_state:
            raise errors.NotMatchDataStructState(origin=ct.origin)
        if not sk.ntt_state or not sk.montgomery_state:
            raise errors.NotMatchDataStructState(origin=sk.origin)

    def multiparty_decrypt_partial(self, ct: data_struct, sk: data_struct):
        """c1 * s on GPU 0's rows, coefficient domain, lazy (eng.py:2472-2493)."""
        self._check_partial_decrypt(ct, sk)
        if 0 not in self.local_ids:
            return None
        level = ct.level
        a = ct.data[1][0].clone()
        self.ntt.enter_ntt([a], level)
        sa = self.ntt.mont_mult([a], [sk.data[0][self.ntt.starts[level][0]:]], level)
        self.ntt.intt_exit(sa, level)
        return sa

    def multiparty_decrypt_head(self, ct: data_struct, sk: data_struct):
        sa = self.multiparty_decrypt_partial(ct, sk)
        return None if sa is None else self.ntt.mont_add([ct.data[0][0]], sa, ct.level)

    def multiparty_decrypt_fusion(self, pcts: list, level=0, include_special=False):
        pt = [x.clone() for x in pcts[0]]
        for pct in pcts[1:]:
            pt = self.ntt.mont_add(pt, pct, level)
        self.ntt.reduce_2q(pt, level)
        base_at = -self.ctx.num_special_primes - 1 if include_special else -1
        self._need_final_scalar(level)
        scaled = self.ntt.mont_sub([pt[0][base_at][None, :]], [pt[0][0][None, :]], -1)
        self.ntt.mont_enter_scalar(scaled, [self.final_scalar[level]], -1)
        self.ntt.reduce_2q(scaled, -1)
        self.ntt.make_signed(scaled, -1)
        return self.decode(m=scaled, level=level)

    def multiparty_create_key_switching_key(self, sk_src: data_struct, sk_dst: data_struct, a=None) -> data_struct:
        return self.create_key_switching_key(sk_src, sk_dst, a=a)

    def multiparty_create_rotation_key(self, sk: data_struct, delta: int, a=None) -> data_struct:
        return self.create_rotation_key(sk, delta, a=a)

    def _sum_key_components(self, keys, components):
        """Part-wise lazy sums of the listed components (0 = b, 1 = a) of key-switch keys, on every GPU.
        (The reference's rotation / galois variants update GPU 0 only, eng.py:2589-2595, 2638-2648 — a slip
        that breaks multi-GPU collective keys; the evk variants, eng.py:2654-2688, update every GPU.)"""
        total = self.clone(keys[0])
        for key in keys[1:]:
            for part_sum, part in zip(total.data, key.data):
                for comp in components:
                    summed = self.ntt.mont_add(part_sum.data[comp], part.data[comp], 0, -2)
                    for dst, src in zip(part_sum.data[comp], summed):
                        dst.copy_(src)
        return total

    def multiparty_generate_rotation_key(self, rotks: list[data_struct]) -> data_struct:
        return self._sum_key_components(rotks, (0,))

    def generate_rotation_crs(self, rotk: data_struct):
        if types.origins["rotk"] not in rotk.origin and types.origins["ksk"] != rotk.origin:
            raise errors.NotMatchType(origin=rotk.origin, to=types.origins["ksk"])
        return [ksk.data[1] for ksk in rotk.data]

    def generate_galois_crs(self, galk: data_struct):
        if galk.origin != types.origins["galk"]:
            raise errors.NotMatchType(origin=galk.origin, to=types.origins["galk"])
        return [[ksk.data[1] for ksk in rotk.data] for rotk in galk.data]

    def multiparty_create_galois_key(self, sk: data_struct, a: list) -> data_struct:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        parts = [self.multiparty_create_rotation_key(sk, delta, a=a[i]) for i, delta in enumerate(self.galois_deltas)]
        return self._new(parts, types.origins["galk"], include_special=True, ntt_state=True, montgomery_state=True)

    def multiparty_generate_galois_key(self, galks: list[data_struct]) -> data_struct:
        first = galks[0]
        return first._replace(data=[self._sum_key_components([g.data[i] for g in galks], (0,))
                                    for i in range(len(first.data))])

    def multiparty_sum_evk_share(self, evks_share: list[data_struct]):
        return self._sum_key_components(evks_share, (0,))

    def multiparty_mult_evk_share_sum(self, evk_sum: data_struct, sk: data_struct):
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        out = self.clone(evk_sum)
        for part in out.data:
            for comp in (0, 1):
                prod = self.ntt.mont_mult(part.data[comp], sk.data, 0, -2)
                for dst, src in zip(part.data[comp], prod):
                    dst.copy_(src)
        return out

    def multiparty_sum_evk_share_mult(self, evk_sum_mult: list[data_struct]) -> data_struct:
        return self._sum_key_components(evk_sum_mult, (0, 1))
