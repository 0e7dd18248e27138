"""Per-GPU NTT / Montgomery parameter packs.

Host-side mirror of the reference's `ntt_context` (src/liberate/ntt/ntt_context.py:14-599): same
attributes (`p, starts, stops, qlists, parts_pack, *_prepack`, the per-device constant lists) and the
same 17 wrapper methods with the `(a, lvl, mult_type, part)` convention, so engine-level code indexes
it exactly as it indexes the reference's.

Index convention of every `*_prepack[mult_type][lvl][part]` (ntt_context.py:417-526):
  mult_type in 0..D-1 : single-device packs of GPU `mult_type`; `part` walks that GPU's key-switch
                        digits at `lvl`, and part = -2 / -1 is that GPU's whole with-special /
                        ordinary row range;
  mult_type in {-2,-1}: all-device packs over the with-special / ordinary rows, `part` = 0.
Each pack is a list of per-parameter lists with one tensor per participating GPU (GPUs whose row
range is empty at that level are dropped, ntt_context.py:191-225).

Differences from the reference, all on the inside:
  * twiddles are compact [rows, N] tables (psi_br / ipsi_br, Montgomery-entered on device with the
    same REDC as the reference's psi_enter, ntt_context.py:115-130), 512 KiB per limb at logN 16
    instead of 8 MiB; `even/odd` gather tables do not exist (entries are None);
  * parameter packs are views built by one generic slicer.
"""
from __future__ import annotations

import datetime
import time

import numpy as np
import torch

from ..fhe.presets import errors
from . import ntt_cuda
from .rns_partition import rns_partition


class ntt_context:
    @errors.log_error
    def __init__(self, ctx, index_type=torch.int32, devices=None, verbose=False, ops=None, local_ids=None, balance=False):
        """`balance`: rns_partition(balance=True) — the top digit on GPU 1 instead of GPU 0 (not the reference's layout).
        `ops`: module/object exposing the 15 `ntt_cuda` functions (default: the HIP shim).
        `local_ids`: logical device ids whose rows this process materialises (default: all) — with one
        process per GPU each rank passes its own id and the other devices' constant tensors stay empty."""
        t0 = time.time()
        self.ops = ntt_cuda if ops is None else ops
        if devices is None:
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        self.devices = [f"cuda:{d}" if isinstance(d, int) else d for d in devices]
        if len(self.devices) == 0:
            raise RuntimeError("ntt_context: no GPU device given/visible; the HIP path has no CPU fallback")
        self.num_devices = len(self.devices)
        self.local_ids = list(range(self.num_devices)) if local_ids is None else list(local_ids)
        self.index_type = index_type
        self.verbose = verbose
        self.ctx = ctx

        self.num_ordinary_primes = ctx.num_scales + 1
        self.num_special_primes = ctx.num_special_primes
        self.num_levels = ctx.num_scales + 1
        self.p = rns_partition(self.num_ordinary_primes, self.num_special_primes, self.num_devices, balance=balance)
        self._say(f"partitioning done: {self.num_levels} levels, {self.num_ordinary_primes} ordinary "
                  f"and {self.num_special_primes} special primes")

        self.prepare_parameters()
        self._say(f"ntt parameters ready after {time.time() - t0:.2f} s")

        self.qlists = [qi.tolist() for qi in self.q]
        self.starts = self.p.diff
        self.stops = [
            [len(d) for d in self.p.destination_arrays_with_special[0]],
            [len(d) for d in self.p.destination_arrays[0]],
        ]
        self.generate_parts_pack()
        self.pre_package()

    def _say(self, msg):
        if self.verbose:
            print(f"[{datetime.datetime.now()}] {msg}")

    # ---------------------------------------------------------------------------------------------
    # constants -> devices
    # ---------------------------------------------------------------------------------------------
    def partition_variable(self, variable):
        """Rows of `variable` (one per prime) gathered per GPU in that GPU's level-0 row order."""
        v = np.asarray(variable, dtype=getattr(self.ctx, "numpy_dtype", np.int64))   # int32 in the reference's 30-bit word mode
        out = []
        for dev_id, (rows, dev) in enumerate(zip(self.p.d_special, self.devices)):
            rows = rows if dev_id in self.local_ids else []
            out.append(torch.from_numpy(np.ascontiguousarray(v[rows])).to(dev))
        return out

    def prepare_parameters(self):
        c = self.ctx
        scale = 1 << c.scale_bits
        self.Rs_scale = self.partition_variable([rs * scale % q for rs, q in zip(c.R_square, c.q)])
        self.Rs = self.partition_variable(c.R_square)
        self.q = self.partition_variable(c.q)
        self._2q = self.partition_variable(c.q_double)
        self.ql = self.partition_variable(c.q_lower_bits)
        self.qh = self.partition_variable(c.q_higher_bits)
        self.kl = self.partition_variable(c.k_lower_bits)
        self.kh = self.partition_variable(c.k_higher_bits)
        self.Ninv = self.partition_variable([ni * c.R % q for ni, q in zip(c.N_inv, c.q)])

        # no gather tables: the kernels compute butterfly indices
        self.even = self.odd = self.ieven = self.iodd = [None] * self.num_devices

        # compact twiddle tables, entered into Montgomery form on device
        self.psi = self.partition_variable(c.psi_br)
        self.ipsi = self.partition_variable(c.ipsi_br)
        live = lambda xs: [xs[i] for i in self.local_ids]
        self.ops.mont_enter(live(self.psi), live(self.Rs), live(self.ql), live(self.qh), live(self.kl), live(self.kh))
        self.ops.mont_enter(live(self.ipsi), live(self.Rs), live(self.ql), live(self.qh), live(self.kl), live(self.kh))

        self.mont_pack0 = [self.ql, self.qh, self.kl, self.kh]
        self.ntt_pack0 = [self.even, self.odd, self.psi, self._2q, self.ql, self.qh, self.kl, self.kh]
        self.intt_pack0 = [self.ieven, self.iodd, self.ipsi, self.Ninv, self._2q, self.ql, self.qh, self.kl, self.kh]

    # ---------------------------------------------------------------------------------------------
    # slicing
    # ---------------------------------------------------------------------------------------------
    def param_pack(self, param, astart, astop, remove_empty=True):
        pack = [param[dev][astart[dev]:astop[dev]] for dev in range(self.num_devices)]
        return [x for x in pack if len(x) > 0] if remove_empty else pack

    def mont_pack(self, astart, astop, remove_empty=True):
        return [self.param_pack(x, astart, astop, remove_empty) for x in self.mont_pack0]

    def _transform_pack(self, pack0, astart, astop, remove_empty):
        rest = [self.param_pack(x, astart, astop, remove_empty=False) for x in pack0[2:]]
        tables = [list(tab) for tab in pack0[:2]]
        if remove_empty:
            keep = [dev for dev in range(self.num_devices) if len(rest[0][dev]) > 0]
            tables = [[tab[dev] for dev in keep] for tab in tables]
            rest = [[x[dev] for dev in keep] for x in rest]
        return tables + rest

    def ntt_pack(self, astart, astop, remove_empty=True):
        return self._transform_pack(self.ntt_pack0, astart, astop, remove_empty)

    def intt_pack(self, astart, astop, remove_empty=True):
        return self._transform_pack(self.intt_pack0, astart, astop, remove_empty)

    def start_stop(self, lvl, mult_type):
        return self.starts[lvl], self.stops[mult_type]

    def params_pack_device(self, device_id, astart, astop):
        """Packs restricted to rows [astart, astop] (inclusive) of one GPU."""
        starts = [0] * self.num_devices
        stops = [0] * self.num_devices
        starts[device_id], stops[device_id] = astart, astop + 1
        return {
            "mont_pack": self.mont_pack(starts, stops),
            "ntt_pack": self.ntt_pack(starts, stops),
            "intt_pack": self.intt_pack(starts, stops),
            "Rs": self.param_pack(self.Rs, starts, stops),
            "Rs_scale": self.param_pack(self.Rs_scale, starts, stops),
            "_2q": self.param_pack(self._2q, starts, stops),
            "qlist": self.param_pack(self.qlists, starts, stops),
        }

    # ---------------------------------------------------------------------------------------------
    # per-digit packs and base-conversion constants (ntt_context.py:253-412)
    # ---------------------------------------------------------------------------------------------
    def digit_constants(self, primes_idx):
        """Mixed-radix constants of one key-switch digit with moduli m_0..m_{a-1} (L_i = m_0...m_i):
        Y_scalar[i] = L_i^-1 * R mod m_{i+1};  L_scalar[i][j-(i+2)] = L_i * R mod m_j, j >= i+2;
        L_enter[dev][i][row] = L_i * R^2 mod q_row over GPU dev's level-0 rows."""
        c = self.ctx
        m = [c.q[i] for i in primes_idx]
        alpha = len(m)
        Ls = []
        for i in range(alpha - 1):
            Ls.append(m[i] if i == 0 else Ls[-1] * m[i])
        Y_scalar = [pow(Ls[i], -1, m[i + 1]) * c.R % m[i + 1] for i in range(alpha - 1)]
        L_scalar = [[Ls[i] * c.R % m[j] for j in range(i + 2, alpha)] for i in range(alpha - 2)]
        L_enter = []
        for dev in range(self.num_devices):
            rows = self.p.destination_arrays_with_special[0][dev]
            L_enter.append([[Ls[i] * c.R_square[r] % c.q[r] for r in rows] for i in range(alpha - 1)])
        return Y_scalar, L_scalar, L_enter

    def generate_parts_pack(self):
        D = self.num_devices
        self.parts_pack = []
        for dev in range(D):
            packs = {}
            for i in range(len(self.p.destination_arrays_with_special[0][dev])):
                packs[(i,)] = self.params_pack_device(dev, i, i)
            for lvl in range(self.num_levels):
                for mult_type in (-1, -2):
                    starts, stops = self.start_stop(lvl, mult_type)
                    key = tuple(range(starts[dev], stops[dev]))
                    if key and key not in packs:
                        packs[key] = self.params_pack_device(dev, key[0], key[-1])
                for part in self.p.p_special[lvl][dev]:
                    key = tuple(part)
                    if key not in packs:
                        packs[key] = self.params_pack_device(dev, key[0], key[-1])
            self.parts_pack.append(packs)

        t64 = lambda x, dev: torch.tensor(x, dtype=torch.int64, device=self.devices[dev])
        for dev in range(D):
            for lvl in range(self.num_levels):
                for part_index, primes_idx in enumerate(self.p.destination_parts[lvl][dev]):
                    item = self.parts_pack[dev][tuple(self.p.p[lvl][dev][part_index])]
                    if "Y_scalar" in item:
                        continue
                    Y_scalar, L_scalar, L_enter = self.digit_constants(primes_idx)
                    if Y_scalar:
                        item["Y_scalar"] = t64(Y_scalar, dev)
                        item["L_enter"] = [[t64(Li, tdev) for Li in L_enter[tdev]] if tdev in self.local_ids else None
                                           for tdev in range(D)]
                    else:
                        item["Y_scalar"] = None
                        item["L_enter"] = [None] * D
                    item["L_scalar"] = [t64(Li, dev) for Li in L_scalar] if L_scalar else None

    # ---------------------------------------------------------------------------------------------
    # [mult_type][lvl][part] lookup tables (ntt_context.py:417-526)
    # ---------------------------------------------------------------------------------------------
    _PREPACK_FIELDS = (
        ("mont_prepack", "mont_pack"), ("ntt_prepack", "ntt_pack"), ("intt_prepack", "intt_pack"),
        ("Rs_prepack", "Rs"), ("Rs_scale_prepack", "Rs_scale"), ("_2q_prepack", "_2q"), ("q_prepack", "qlist"),
    )

    def pre_package(self):
        for attr, _ in self._PREPACK_FIELDS:
            setattr(self, attr, [])
        # single-device packs
        for dev in range(self.num_devices):
            per_field = {attr: [] for attr, _ in self._PREPACK_FIELDS}
            for lvl in range(self.num_levels):
                items = [self.parts_pack[dev][tuple(part)] for part in self.p.p_special[lvl][dev]]
                for mult_type in (-2, -1):
                    starts, stops = self.start_stop(lvl, mult_type)
                    key = tuple(range(starts[dev], stops[dev]))
                    items.append(self.parts_pack[dev][key] if key else None)
                for attr, field in self._PREPACK_FIELDS:
                    per_field[attr].append([None if it is None else it[field] for it in items])
            for attr, _ in self._PREPACK_FIELDS:
                getattr(self, attr).append(per_field[attr])
        # all-device packs: index -2 = with special rows, -1 = ordinary rows
        for mult_type in (-2, -1):
            per_field = {attr: [] for attr, _ in self._PREPACK_FIELDS}
            for lvl in range(self.num_levels):
                stst = self.start_stop(lvl, mult_type)
                per_field["mont_prepack"].append([self.mont_pack(*stst)])
                per_field["ntt_prepack"].append([self.ntt_pack(*stst)])
                per_field["intt_prepack"].append([self.intt_pack(*stst)])
                per_field["Rs_prepack"].append([self.param_pack(self.Rs, *stst)])
                per_field["Rs_scale_prepack"].append([self.param_pack(self.Rs_scale, *stst)])
                per_field["_2q_prepack"].append([self.param_pack(self._2q, *stst)])
                per_field["q_prepack"].append([self.param_pack(self.qlists, *stst)])
            for attr, _ in self._PREPACK_FIELDS:
                getattr(self, attr).append(per_field[attr])

    # ---------------------------------------------------------------------------------------------
    # op wrappers (ntt_context.py:532-599)
    # ---------------------------------------------------------------------------------------------
    def mont_enter(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.mont_enter(a, self.Rs_prepack[mult_type][lvl][part], *self.mont_prepack[mult_type][lvl][part])

    def mont_enter_scale(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.mont_enter(a, self.Rs_scale_prepack[mult_type][lvl][part], *self.mont_prepack[mult_type][lvl][part])

    def mont_enter_scalar(self, a, b, lvl=0, mult_type=-1, part=0):
        self.ops.mont_enter(a, b, *self.mont_prepack[mult_type][lvl][part])

    def mont_mult(self, a, b, lvl=0, mult_type=-1, part=0):
        return self.ops.mont_mult(a, b, *self.mont_prepack[mult_type][lvl][part])

    def ntt(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.ntt(a, *self.ntt_prepack[mult_type][lvl][part])

    def enter_ntt(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.enter_ntt(a, self.Rs_prepack[mult_type][lvl][part], *self.ntt_prepack[mult_type][lvl][part])

    def intt(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.intt(a, *self.intt_prepack[mult_type][lvl][part])

    def mont_redc(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.mont_redc(a, *self.mont_prepack[mult_type][lvl][part])

    def intt_exit(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.intt_exit(a, *self.intt_prepack[mult_type][lvl][part])

    def intt_exit_reduce(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.intt_exit_reduce(a, *self.intt_prepack[mult_type][lvl][part])

    def intt_exit_reduce_signed(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.intt_exit_reduce_signed(a, *self.intt_prepack[mult_type][lvl][part])

    def reduce_2q(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.reduce_2q(a, self._2q_prepack[mult_type][lvl][part])

    def make_signed(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.make_signed(a, self._2q_prepack[mult_type][lvl][part])

    def make_unsigned(self, a, lvl=0, mult_type=-1, part=0):
        self.ops.make_unsigned(a, self._2q_prepack[mult_type][lvl][part])

    def mont_add(self, a, b, lvl=0, mult_type=-1, part=0):
        return self.ops.mont_add(a, b, self._2q_prepack[mult_type][lvl][part])

    def mont_sub(self, a, b, lvl=0, mult_type=-1, part=0):
        return self.ops.mont_sub(a, b, self._2q_prepack[mult_type][lvl][part])

    def tile_unsigned(self, a, lvl=0, mult_type=-1, part=0):
        return self.ops.tile_unsigned(a, self._2q_prepack[mult_type][lvl][part])
