"""`ckks_engine`: the RNS-CKKS evaluator, API mirror of the reference's src/liberate/fhe/ckks_engine.py.

Same constructor, method names, argument meaning, state checks and exceptions as the reference class
(`eng.py` below = that file), so code written against `liberate.fhe.ckks_engine` runs against this
one.  What is different is underneath:

  * the hot methods — `rescale`, `cc_mult`, `relinearize`, `create_switcher` / `switch_key`,
    `rotate_single`, `conjugate` — do not replay the reference's ~100 tiny launches per key switch;
    each is a handful of fused HIP kernels (include/ckks_hip.h, "engine-level fused ops") whose
    per-word arithmetic is op-for-op the reference's, so results are bit-identical;
  * key-switch keys live in ONE packed tensor per GPU ([parts, 2, rows, N]); the `data_struct`s the
    reference API exposes are views into it, so the inner product streams the key with plain strides;
  * RNS limbs shard over GPUs by the reference's `rns_partition`, but the two exchange steps
    (rescale row broadcast eng.py:999-1011, key-switch digit gather eng.py:778-810) go through a
    `comm` object — RCCL broadcast / all-gather over xGMI with one process per GPU — instead of
    pinned-host staging.  With `comm=None` one process drives every listed device, as the reference does.

Data lists (`data_struct.data[i]`) hold one tensor per LOCAL participating device, ordered by device id
— identical to the reference's layout when one process owns all devices.
"""
from __future__ import annotations

import ctypes
import math
import os
import weakref
from hashlib import sha256

import numpy as np
import torch

from ..csprng import Csprng
from ..ntt import ntt_context
from . import encdec
from .backend import Consts
from .context.ckks_context import ckks_context
from .data_struct import data_struct
from .evaluator import EvaluatorOps, is_struct
from .presets import errors, types
from .version import VERSION


class _OneShard:
    """The ntt_context calls of a decryption restricted to ONE local shard (pack index `li`, logical device `dev`): the
    context's wrappers walk every device's list from index 0, which is what the reference's decryption — device 0 only — wants;
    the balanced limb map also needs the first row of device 1."""

    def __init__(self, eng, li, dev):
        self.n, self.li, self.dev, self.starts = eng.ntt, li, dev, eng.ntt.starts

    def _pk(self, pack):
        return [x[self.li:self.li + 1] for x in pack]

    def enter_ntt(self, a, lvl):
        self.n.ops.enter_ntt(a, self.n.Rs_prepack[-1][lvl][0][self.li:self.li + 1], *self._pk(self.n.ntt_prepack[-1][lvl][0]))

    def mont_mult(self, a, b, lvl):
        return self.n.ops.mont_mult(a, b, *self._pk(self.n.mont_prepack[-1][lvl][0]))

    def intt_exit(self, a, lvl):
        self.n.ops.intt_exit(a, *self._pk(self.n.intt_prepack[-1][lvl][0]))

    def intt_exit_reduce(self, a, lvl):
        self.n.ops.intt_exit_reduce(a, *self._pk(self.n.intt_prepack[-1][lvl][0]))

    def mont_add(self, a, b, lvl):
        return self.n.ops.mont_add(a, b, self.n._2q_prepack[-1][lvl][0][self.li:self.li + 1])

    def reduce_2q(self, a, lvl):
        self.n.ops.reduce_2q(a, self.n._2q_prepack[-1][lvl][0][self.li:self.li + 1])

    def _decrypt_rows(self, ct, sk):
        return ckks_engine._decrypt_rows_on(self, ct, sk, self.li, self.dev)


class ckks_engine(EvaluatorOps):
    @errors.log_error
    def __init__(self, devices: list[int] = None, verbose: bool = False, bias_guard: bool = True,
                 norm: str = "forward", backend=None, comm=None, balanced_limb_map: bool = False, **ctx_params):
        if backend is None:
            from .backend import HipBackend  # raises if libckks_hip.so is missing: no fallback
            backend = HipBackend()
        self.backend = backend
        self.comm = comm
        self.bias_guard = bias_guard
        self.norm = norm
        self.version = VERSION
        self.ctx = ckks_context(**ctx_params)
        if self.ctx.buffer_bit_length != 62:
            # the 30-bit word mode is served below the engine: ckks_context, ntt_context and the 15 ntt_cuda functions
            # (csrc/ckks_w30.hip).  The reference's own engine constructs in that mode but cannot generate a key in it — its
            # samplers hand int64 words to kernels dispatched on int32 constants (ckks_engine.py:355) — so there is no
            # reference behaviour to reproduce above the boundary.
            raise ValueError("ckks_engine: the engine's ops exist for buffer_bit_length = 62; the 30-bit word mode is served by "
                             "ckks_context / ntt_context / ntt_cuda only (as far as the reference itself works in it)")

        # one process per GPU: the limb-sharded code path.  A communicator of ONE rank takes the ordinary single-device path unless
        # it says `solo_sharded` (comm.py): then the rank runs the sharded halves and issues both exchange steps to the
        # communicator — the form a one-GPU lease can execute against a real RCCL communicator
        self._multi = comm is not None and (comm.world_size > 1 or bool(getattr(comm, "solo_sharded", False)))
        if self._multi:
            local = devices[0] if devices else comm.local_device
            logical, local_ids = [local] * comm.world_size, [comm.rank]
        else:
            logical, local_ids = devices, None
        # balanced_limb_map (default off: the reference's layout): rns_partition(balance=True) — rank 0 of a limb-sharded gold
        # engine over 8 GPUs holds 5 rows instead of 7 (the largest share 6 instead of 7): same digits, same words per limb
        self.ntt = ntt_context(self.ctx, devices=logical, verbose=verbose, ops=backend.ops, local_ids=local_ids, balance=balanced_limb_map)
        self.local_ids = self.ntt.local_ids
        self.num_levels = self.ntt.num_levels - 1
        self.num_slots = self.ctx.N // 2

        # Every rank runs the same ChaCha20 key / nonce; the counters make the streams distinct (csprng.py:216-223).
        seed_words = None
        if self._multi:
            fresh = torch.tensor([int.from_bytes(os.urandom(4), "big") for _ in range(10)], dtype=torch.int64)
            seed_words = comm.broadcast(fresh.to(comm.local_device), src=0, shape=(10,), device=comm.local_device).tolist()
        self.rng = getattr(backend, "csprng_class", Csprng)(self.ctx.N, [len(di) for di in self.ntt.p.d], max(self.ntt.num_special_primes, 2),
                          devices=self.ntt.devices, local_ids=self.local_ids,
                          seed=seed_words[:8] if seed_words else None, nonce=seed_words[8:] if seed_words else None)

        self.int_scale = 2 ** self.ctx.scale_bits
        self.scale = np.float64(self.int_scale)
        qstr = ",".join(str(qi) for qi in self.ctx.q)
        self.hash = sha256((self.ctx.generation_string + "_" + qstr).encode("utf-8")).hexdigest()

        self.make_adjustments_and_corrections()
        self.device0 = self.ntt.devices[0]
        self.make_mont_PR()
        self.create_ksk_rescales()
        self.alloc_parts()
        self.leveled_devices()
        self.create_rescale_scales()
        self.galois_deltas = [2 ** i for i in range(self.ctx.logN - 1)]
        self._tables = {}
        self._key_packs = {}
        self._workspace = {}
        self._md_ready = set()
        self._lane = 0          # pipeline lane whose workspaces / stream the batched ops currently use
        # a rank of a limb-sharded op replays its fixed-address launches from HIP graphs (see _sharded_segments); LF_ENGINE_GRAPHS=0
        # keeps every launch eager
        self.graph_sharded = os.environ.get("LF_ENGINE_GRAPHS", "1") != "0"
        # the column-form extension of the key switch in Horner form (one product per word fewer; _ks_tables); False keeps the
        # sum over L_{i-1} y_i — same residues either way (tools/ab_engines.py times one against the other)
        self.ks_horner = True
        self._lane_streams = {}
        self._last_stream = {}  # (device, lane) -> the torch stream its last op was enqueued on (_same_stream)
        self._check_kernel_limits()

        ds, nd, ls = data_struct, np.ndarray, list
        self.mult_dispatch_dict = {
            (ds, ds): self.auto_cc_mult, (ls, ds): self.mc_mult, (nd, ds): self.mc_mult, (ds, nd): self.cm_mult,
            (ds, ls): self.cm_mult, (float, ds): self.scalar_mult, (ds, float): self.mult_scalar,
            (int, ds): self.int_scalar_mult, (ds, int): self.mult_int_scalar}
        self.add_dispatch_dict = {
            (ds, ds): self.auto_cc_add, (ls, ds): self.mc_add, (nd, ds): self.mc_add, (ds, nd): self.cm_add,
            (ds, ls): self.cm_add, (float, ds): self.scalar_add, (ds, float): self.add_scalar,
            (int, ds): self.scalar_add, (ds, int): self.add_scalar}
        self.sub_dispatch_dict = {
            (ds, ds): self.auto_cc_sub, (ls, ds): self.mc_sub, (nd, ds): self.mc_sub, (ds, nd): self.cm_sub,
            (ds, ls): self.cm_sub, (float, ds): self.scalar_sub, (ds, float): self.sub_scalar,
            (int, ds): self.scalar_sub, (ds, int): self.sub_scalar}

    # =============================================================================================
    # small helpers
    # =============================================================================================
    def _t64(self, values, dev_id):
        return torch.tensor(values, dtype=torch.int64, device=self.ntt.devices[dev_id])

    def _loc(self, level, special=False):
        """Local device ids that hold rows at `level` (ordinary rows unless special=True)."""
        alive = range(self.ntt.num_devices) if special else range(self.len_devices[level])
        return [d for d in alive if d in self.local_ids]

    def _consts(self, dev, level, special):
        """Montgomery constants of device `dev`'s row range at `level`."""
        key = ("consts", dev, level, special)
        if key not in self._tables:
            a, b = self.ntt.starts[level][dev], self.ntt.stops[0 if special else 1][dev]
            n = self.ntt
            primes = np.array([self.ctx.q[i] for i in n.p.d_special[dev][a:b]], dtype=np.int64)
            self._tables[key] = Consts(n.ql[dev][a:b], n.qh[dev][a:b], n.kl[dev][a:b], n.kh[dev][a:b], n._2q[dev][a:b],
                                       q_host=primes)
        return self._tables[key]

    def _rows(self, dev, level, special):
        return self.ntt.stops[0 if special else 1][dev] - self.ntt.starts[level][dev]

    def _tw(self, dev, level, special, inverse=False):
        """Twiddle rows of device `dev`'s limbs at `level` (one view object per key: the backend keys its per-table
        lookups on it)."""
        key = ("tw", dev, level, special, inverse)
        t = self._tables.get(key)
        if t is None:
            a, b = self.ntt.starts[level][dev], self.ntt.stops[0 if special else 1][dev]
            t = self._tables[key] = (self.ntt.ipsi if inverse else self.ntt.psi)[dev][a:b]
        return t

    def _vec(self, name, dev, level, special):
        key = ("vec", name, dev, level, special)
        t = self._tables.get(key)
        if t is None:
            a, b = self.ntt.starts[level][dev], self.ntt.stops[0 if special else 1][dev]
            t = self._tables[key] = getattr(self.ntt, name)[dev][a:b]
        return t

    def _moddown_ws(self, key, count, ell, K, d, tabs, cs):
        """(workspace, one_launch) of a mod-down of `count` polynomials on device d: with up to backend.moddown_one_max_K
        special primes the mod-down is ONE launch that only reads the workspace's level constants, written here once."""
        N = self.ctx.N
        ws = self._ws(key, (self.backend.moddown_ws_words(count, ell, K, N),), d)
        one = K <= getattr(self.backend, "moddown_one_max_K", 0) and tabs[("pip", d)] is not None
        if one and id(ws) not in self._md_ready:
            self.backend.moddown_consts(ws, count, ell, K, N, tabs[("pip", d)], cs)
            self._md_ready.add(id(ws))      # (workspaces live as long as the engine: the id stays theirs)
        return ws, one

    def _same_stream(self, dev_id):
        """The scratch of (device, lane) is ordered by the stream its ops run on.  An op enqueued on ANOTHER current stream than the
        lane's previous op first makes that stream wait for the previous one (a device-side dependency, no host block): a caller
        that alternates streams between ops of one engine gets serialised scratch instead of a race.  One pointer compare per op."""
        dev = self.ntt.devices[dev_id]
        if not str(dev).startswith("cuda"):
            return
        cur = torch.cuda.current_stream(dev)
        k = (dev_id, self._lane)
        last = self._last_stream.get(k)
        if last is not None and last.cuda_stream != cur.cuda_stream:
            cur.wait_stream(last)
        self._last_stream[k] = cur

    def _ws(self, key, shape, dev_id):
        """Reusable scratch tensor (never returned to the caller); one set per pipeline lane (see _lanes).
        STREAMS: the scratch belongs to the engine, not to a stream — the reference allocates its intermediates per call, this
        engine reuses them, ordered by the stream its ops are enqueued on.  Ops of one engine on ONE torch stream per device are
        ordered by that stream; an op that arrives on another current stream is made to wait for the lane's previous op
        (_same_stream): alternating streams is safe and serial.  For concurrency use one engine per stream — ops of DIFFERENT
        engines are independent (the batched methods fork and join their own lanes)."""
        self._same_stream(dev_id)
        k = (key, tuple(shape), dev_id, self._lane)
        t = self._workspace.get(k)
        if t is None:
            t = torch.empty(shape, dtype=torch.int64, device=self.ntt.devices[dev_id])
            self._workspace[k] = t
        return t

    # =============================================================================================
    # pre-calculations (eng.py:123-263)
    # =============================================================================================
    def create_rescale_scales(self):
        """rescale_scales[level][dev][i] = q_level^-1 * R mod q_i over the rows that survive (eng.py:123-146)."""
        self.rescale_scales = []
        for level in range(self.num_levels):
            per_dev = []
            m0 = self.ctx.q[level]
            for dev, dest in enumerate(self.ntt.p.destination_arrays[level]):
                rows = dest[1:] if dev == self.ntt.p.rescaler_loc[level] else dest
                vals = [pow(m0, -1, self.ctx.q[i]) * self.ctx.R % self.ctx.q[i] for i in rows]
                per_dev.append(self._t64(vals if dev in self.local_ids else [], dev))
            self.rescale_scales.append(per_dev)

    def leveled_devices(self):
        self.len_devices = [len([a for a in self.ntt.p.p[level] if len(a) > 0]) for level in range(self.num_levels)]
        self.neighbor_devices = [
            [[d for d in range(n) if d != src] for src in range(n)] for n in self.len_devices
        ]

    def alloc_parts(self):
        self.parts_alloc = []
        for level in range(self.num_levels):
            counts = [len(parts) for parts in self.ntt.p.p[level]]
            self.parts_alloc.append(
                [alloc[-counts[d] - 1:-1] for d, alloc in enumerate(self.ntt.p.part_allocations)])
        self.stor_ids = []
        for level in range(self.num_levels):
            alloc = self.parts_alloc[level]
            min_id = min(min(a) for a in alloc if len(a) > 0)
            self.stor_ids.append([[i - min_id for i in a] for a in alloc])

    def create_ksk_rescales(self):
        """PiRs[level][P_ind][dev][row] = P_j^-1 * R mod q_row, specials last-first (eng.py:183-216)."""
        R, K = self.ctx.R, self.ntt.num_special_primes
        P = self.ctx.q[-K:][::-1]
        self.PiRs = [[]]
        for P_ind, Pj in enumerate(P):
            per_dev = []
            for dev in range(self.ntt.num_devices):
                dest = self.ntt.p.destination_arrays_with_special[0][dev]
                vals = [pow(Pj, -1, self.ctx.q[i]) * R % self.ctx.q[i] for i in dest[:-P_ind - 1]]
                per_dev.append(self._t64(vals if dev in self.local_ids else [], dev))
            self.PiRs[0].append(per_dev)
        for level in range(1, self.num_levels):
            self.PiRs.append([[self.PiRs[0][P_ind][dev][self.ntt.starts[level][dev]:]
                               for dev in range(self.ntt.num_devices)] for P_ind in range(K)])

    def make_mont_PR(self):
        P = math.prod(self.ctx.q[-self.ntt.num_special_primes:])
        PR = P * self.ctx.R
        self.mont_PR = []
        for dev in range(self.ntt.num_devices):
            dest = self.ntt.p.destination_arrays[0][dev] if dev in self.local_ids else []
            self.mont_PR.append(self._t64([PR % self.ctx.q[i] for i in dest], dev))

    def make_adjustments_and_corrections(self):
        q, ns = self.ctx.q, self.ctx.num_scales
        self.alpha = [(self.scale / np.float64(qi)) ** 2 for qi in q[:ns]]
        self.deviations = [1]
        for al in self.alpha:
            self.deviations.append(self.deviations[-1] ** 2 * al)
        # The final scaling of a decryption reads a scale-prime row ("scaler") beside the base-prime row.  In the reference's
        # limb map both sit on device 0 at every level (eng.py:517-533: its first and last row).  In the balanced map device 0
        # holds only the base prime once its scale digit is gone: the scaler is then the first row of device 1, which holds the
        # highest scale primes (_scaler_dev; decrypt fetches that one row).
        base = self.ntt.p.base_prime_idx
        self._scaler_dev = [0 if (da[0][0] != base or len(da) < 2) else 1 for da in self.ntt.p.destination_arrays[:-1]]
        self.final_q_ind = [da[sd][0] for da, sd in zip(self.ntt.p.destination_arrays[:-1], self._scaler_dev)]
        self.final_q = [q[i] for i in self.final_q_ind]
        self.final_alpha = [self.scale / np.float64(x) for x in self.final_q]
        self.corrections = [1 / (d * fa) for d, fa in zip(self.deviations, self.final_alpha)]
        self.base_prime = q[self.ntt.p.base_prime_idx]
        # (a level whose only limb is the base prime has nothing to scale by: None, _need_final_scalar refuses)
        self.final_scalar = [None if x == self.base_prime else self._t64([pow(x, -1, self.base_prime) * self.ctx.R % self.base_prime], 0)
                             for x in self.final_q]

    # =============================================================================================
    # examples, encode / decode (eng.py:269-343)
    # =============================================================================================
    def absmax_error(self, x, y):
        if type(x[0]) == np.complex128 and type(y[0]) == np.complex128:
            return np.abs(x.real - y.real).max() + np.abs(x.imag - y.imag).max() * 1j
        return np.abs(np.array(x) - np.array(y)).max()

    def integral_bits_available(self):
        return math.floor(math.log2(self.base_prime)) - self.ctx.scale_bits

    @errors.log_error
    def example(self, amin=None, amax=None, decimal_places: int = 10) -> np.array:
        if amin is None:
            amin = -(2 ** self.integral_bits_available())
        if amax is None:
            amax = 2 ** self.integral_bits_available()
        base = 10 ** decimal_places
        a = np.random.randint(amin * base, amax * base, self.ctx.N // 2) / base
        b = np.random.randint(amin * base, amax * base, self.ctx.N // 2) / base
        return a + b * 1j

    def padding(self, m):
        try:
            return np.pad(m, (0, self.num_slots - len(m)), constant_values=(0, 0))
        except TypeError:
            return np.pad([m], (0, self.num_slots - 1), constant_values=(0, 0))
        except Exception:
            raise Exception("[Error] encoding Padding Error.")

    def _replicate_plain(self, pt):
        """The encoded polynomial on every local device (reference: pinned-host broadcast, eng.py:327-331).
        Every rank encodes with the shared random stream, so no exchange is needed."""
        return [pt.to(self.ntt.devices[d]) for d in self.local_ids]

    @errors.log_error
    def encode(self, m, level: int = 0, padding=True) -> list[torch.Tensor]:
        if padding:
            m = self.padding(m)
        dev0 = self.ntt.devices[self.local_ids[0]]
        pt = encdec.encode(m, scale=self.scale, rng=self.rng, device=dev0, deviation=self.deviations[level], norm=self.norm)
        return self._replicate_plain(pt)

    @errors.log_error
    def decode(self, m, level=0, is_real: bool = False) -> list:
        decoded = encdec.decode(m[0].squeeze(), scale=self.scale, correction=self.corrections[level], norm=self.norm)
        out = decoded[:self.ctx.N // 2].cpu().numpy()
        return out.real if is_real else out

    # =============================================================================================
    # keys (eng.py:350-411, 601-652, 1054-1070, 1157-1232, 1694-1716)
    # =============================================================================================
    def _new(self, data, origin, level=0, include_special=False, ntt_state=False, montgomery_state=False):
        return data_struct(data=data, include_special=include_special, ntt_state=ntt_state,
                           montgomery_state=montgomery_state, origin=origin, level=level, hash=self.hash,
                           version=self.version)

    def _live(self, xs):
        return [x for x in xs if x is not None]

    @errors.log_error
    def create_secret_key(self, include_special: bool = True) -> data_struct:
        ternary = self._live(self.rng.randint(amax=3, shift=-1, repeats=1))
        mult_type = -2 if include_special else -1
        sk = self.ntt.tile_unsigned(ternary, lvl=0, mult_type=mult_type)
        self.ntt.enter_ntt(sk, 0, mult_type)
        return self._new(sk, types.origins["sk"], include_special=include_special, ntt_state=True, montgomery_state=True)

    @errors.log_error
    def create_public_key(self, sk: data_struct, include_special: bool = False, a: list[torch.Tensor] = None) -> data_struct:
        """pk = (e - a*sk, a) in NTT / Montgomery form (eng.py:369-411)."""
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        if include_special and not sk.include_special:
            raise errors.SecretKeyNotIncludeSpecialPrime()
        mult_type = -2 if include_special else -1
        e = self._live(self.rng.discrete_gaussian(repeats=1))
        e = self.ntt.tile_unsigned(e, 0, mult_type)
        self.ntt.enter_ntt(e, 0, mult_type)
        repeats = self.ctx.num_special_primes if sk.include_special else 0
        if a is None:
            a = self._live(self.rng.randint(self._moduli(mult_type), repeats=repeats))
        sa = self.ntt.mont_mult(a, sk.data, 0, mult_type)
        pk0 = self.ntt.mont_sub(e, sa, 0, mult_type)
        return self._new((pk0, a), types.origins["pk"], include_special=include_special, ntt_state=True, montgomery_state=True)

    def _moduli(self, mult_type):
        """Per-device moduli lists for the uniform sampler (all devices, so the shared stream lines up)."""
        stop = self.ntt.stops[0 if mult_type == -2 else 1]
        return [[self.ctx.q[i] for i in self.ntt.p.d_special[d][:stop[d]]] for d in range(self.ntt.num_devices)]

    def create_key_switching_key(self, sk_from: data_struct, sk_to: data_struct, a=None) -> data_struct:
        """ksk[part] = pk-like (e - a*sk_to + [P*sk_from on that part's rows], a), eng.py:601-652.
        All parts of one device are stored in ONE packed tensor [parts, 2, rows, N]."""
        if sk_from.origin != types.origins["sk"] or sk_to.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin="not a secret key", to=types.origins["sk"])
        if (not sk_from.ntt_state) or (not sk_from.montgomery_state):
            raise errors.NotMatchDataStructState(origin=sk_from.origin)
        if (not sk_to.ntt_state) or (not sk_to.montgomery_state):
            raise errors.NotMatchDataStructState(origin=sk_to.origin)

        p = self.ntt.p
        loc = self._loc(0, special=True)
        stops = self.ntt.stops[-1]
        Psk = [sk_from.data[i][:stops[d]].clone() for i, d in enumerate(loc)]
        self.ntt.mont_enter_scalar(Psk, [self.mont_PR[d] for d in loc], 0)

        nparts = p.num_partitions + 1
        packs = [torch.empty((nparts, 2, self.ntt.stops[0][d], self.ctx.N), dtype=torch.int64,
                             device=self.ntt.devices[d]) for d in loc]
        ksk = [[] for _ in range(nparts)]
        for owner in range(self.ntt.num_devices):
            for part_id, part in enumerate(p.p[0][owner]):
                gid = p.part_allocations[owner][part_id]
                crs = a[gid] if a else None
                pk = self.create_public_key(sk_to, include_special=True, a=crs)
                b_views, a_views = [], []
                for i, d in enumerate(loc):
                    packs[i][gid, 0].copy_(pk.data[0][i])
                    packs[i][gid, 1].copy_(pk.data[1][i])
                    b_views.append(packs[i][gid, 0])
                    a_views.append(packs[i][gid, 1])
                if owner in loc:
                    i = loc.index(owner)
                    lo, hi = part[0], part[-1] + 1
                    shard = b_views[i][lo:hi]
                    _2q = self.ntt.parts_pack[owner][tuple(part)]["_2q"]
                    shard.copy_(self.backend.ops.mont_add([shard], [Psk[i][lo:hi]], _2q)[0])
                ksk[gid] = pk._replace(data=(b_views, a_views), origin=f"key switch key part index {gid}")
        out = self._new(ksk, types.origins["ksk"], include_special=True, ntt_state=True, montgomery_state=True)
        self._remember_pack(out, packs, own=True)
        return out

    def create_evk(self, sk: data_struct) -> data_struct:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        sk2 = self._new(self.ntt.mont_mult(sk.data, sk.data, 0, -2), types.origins["sk"], level=sk.level,
                        include_special=True, ntt_state=True, montgomery_state=True)
        return self.create_key_switching_key(sk2, sk)

    def _permuted_secret(self, sk, exponent):
        """sk(X^p) in NTT / Montgomery form (eng.py:1161-1164, 1701-1704)."""
        loc = self._loc(0, special=True)
        out = []
        for i, d in enumerate(loc):
            t = sk.data[i].clone()
            rows = self._rows(d, 0, special=False)
            c = self._consts(d, 0, False)
            self.backend.intt(t, 1, rows, self.ctx.logN, self._tw(d, 0, False, True), self._vec("Ninv", d, 0, False), 0, c)
            r = torch.zeros_like(t)
            self.backend.galois(t, r, rows, self.ctx.logN, exponent, None)
            self.backend.ntt(r, 1, rows, self.ctx.logN, self._tw(d, 0, False), None, c)
            out.append(r)
        return self._new(out, types.origins["sk"], ntt_state=True, montgomery_state=True)

    def create_rotation_key(self, sk: data_struct, delta: int, a: list[torch.Tensor] = None) -> data_struct:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        sk_rot = self._permuted_secret(sk, encdec.galois_exponent(self.ctx.N, delta))
        rotk = self.create_key_switching_key(sk_rot, sk, a=a)
        return rotk._replace(origin=types.origins["rotk"] + f"{delta}")   # same tensors: same packed key

    def create_galois_key(self, sk: data_struct) -> data_struct:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        parts = [self.create_rotation_key(sk, delta) for delta in self.galois_deltas]
        return self._new(parts, types.origins["galk"], include_special=True, ntt_state=True, montgomery_state=True)

    def create_conjugation_key(self, sk: data_struct) -> data_struct:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        if (not sk.ntt_state) or (not sk.montgomery_state):
            raise errors.NotMatchDataStructState(origin=sk.origin)
        sk_conj = self._permuted_secret(sk, encdec.conjugation_exponent(self.ctx.N))
        k = self.create_key_switching_key(sk_conj, sk)
        return k._replace(origin=types.origins["conjk"])

    # ---- packed-key cache --------------------------------------------------------------------------------------
    # Identity of a key = its first tensor (part 0, component b, first local device): data_structs and lists cannot
    # be weakly referenced, tensors can.  An entry dies with that tensor (weakref.finalize), so a key the caller
    # drops releases its pack (gold: 429 MB per key, and as much again for the planes copy of the fused path) and a
    # recycled id() can never alias a dead entry.
    @staticmethod
    def _key_anchor(ksk):
        return ksk.data[0].data[0][0]

    @staticmethod
    def _key_versions(ksk):
        return tuple(t._version for part in ksk.data for comp in part.data for t in comp)

    def _remember_pack(self, ksk, packs, own):
        anchor = self._key_anchor(ksk)
        key = id(anchor)
        # `own`: the key's tensors ARE views of the pack (in-place edits land in it); a foreign key's pack is a copy,
        # rebuilt when any of its tensors has been modified in place since (torch's version counters)
        self._key_packs[key] = {"ref": weakref.ref(anchor), "packs": packs, "own": own,
                                "versions": None if own else self._key_versions(ksk)}
        weakref.finalize(anchor, self._forget_key, self._key_packs, self._tables, key)
        return packs

    @staticmethod
    def _purge_graphs(tables, tensors):
        """Captured segments of a sharded rank hold the pack they were captured on (its address is baked into the graphs, the
        entry keeps the tensor alive): drop those of `tensors`."""
        ptrs = {t.data_ptr() for t in tensors}
        for k in [k for k in tables if isinstance(k, tuple) and k and k[0] == "sgraph" and k[5] in ptrs]:
            del tables[k]

    @staticmethod
    def _forget_key(key_packs, tables, key):
        """A key's anchor tensor died: its pack, its planes copy and the graphs captured on either go with it."""
        gone = key_packs.pop(key, None)
        if gone:
            ckks_engine._purge_graphs(tables, list(gone.get("packs") or []) + list(gone.get("planes") or []))

    def _planes_wanted(self):
        """Fused key switches (two-pass ring degrees) read the key in the planes format (include/ckks_hip.h LF_KEY_PLANES:
        the inner product with the key runs at the HBM rate and most of its bytes are key words; fp64-class rows shrink to
        12 instead of 16 bytes per word pair).  The unfused path (logN <= 12) and checker backends read raw words."""
        return getattr(self.backend, "key_planes", None) is not None and self.ctx.logN >= self.backend.fused_ks_min_logN

    def _planes_of(self, blocks_of_part, i, d):
        """Planes-format tensor [parts, 2, rows, N] of local device index i from per-part (b rows, a rows) tensors."""
        c = self._consts(d, 0, True)
        nparts = len(blocks_of_part)
        out = torch.empty((nparts, 2, self.ntt.stops[0][d], self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
        for p_, (b_rows, a_rows) in enumerate(blocks_of_part):
            self.backend.key_planes(b_rows.contiguous(), a_rows.contiguous(), out[p_, 0], out[p_, 1], c)
        return self.backend.mark_planes(out)

    def _key_pack(self, ksk):
        """Packed per-device key tensors of a key-switch key, in the format the engine's key switch reads (planes for the
        fused path, raw words otherwise); built once per key, on the CURRENT stream — callers that use the pack from
        another stream order themselves after this call (see _run_groups: it is made before the side lane is forked).
        A key made by this engine is a set of views of its raw pack (in-place edits land there): its planes copy is
        rebuilt when the pack's version counter moves.  A foreign key is converted part by part, no raw copy is kept."""
        anchor = self._key_anchor(ksk)
        hit = self._key_packs.get(id(anchor))
        planes = self._planes_wanted()
        loc = self._loc(0, special=True)
        if hit is not None and hit["ref"]() is anchor:
            if hit["own"]:
                if hit.get("compact"):
                    return hit["planes"]
                if not planes:
                    return hit["packs"]
                ver = tuple(p._version for p in hit["packs"])
                if hit.get("planes_ver") != ver:
                    self._purge_graphs(self._tables, hit.get("planes") or [])     # graphs captured on the copy being replaced
                    hit["planes"] = [self._planes_of([(pk[q_, 0], pk[q_, 1]) for q_ in range(pk.size(0))], i, d)
                                     for i, (d, pk) in enumerate(zip(loc, hit["packs"]))]
                    hit["planes_ver"] = ver
                return hit["planes"]
            if hit["versions"] == self._key_versions(ksk):
                return hit["packs"]
        packs = []
        for i, d in enumerate(loc):
            if planes:
                packs.append(self._planes_of([(part.data[0][i], part.data[1][i]) for part in ksk.data], i, d))
            else:
                parts = [torch.stack([part.data[0][i], part.data[1][i]]) for part in ksk.data]
                packs.append(torch.stack(parts).contiguous())
        return self._remember_pack(ksk, packs, own=False)

    # ---- keys at half the memory ------------------------------------------------------------------------------------
    # A key made by this engine is a raw pack [parts, 2, rows, N] (its data_structs are views of it: the reference's layout,
    # eng.py:601-652) AND, once a fused key switch has used it, a planes copy of the same size (gold: 2 x 429 MB; a Galois set of
    # 15 rotation keys ~ 13 GB).  The fused path only ever reads the planes.  compact_key() frees the raw pack's STORAGE (the
    # tensors and every view stay valid objects of zero-size storage), expand_key() brings the words back from the planes.
    def compact_key(self, ksk):
        """Free the raw words of a key this engine made; the fused key switch goes on reading its planes copy (gold: 858 -> 429 MB
        per key).  Until expand_key() the key's own tensors hold no data: reading them (save, the unfused path of logN <= 12,
        another engine) is an error raised by torch.  Returns the bytes freed.  In-place edits of a compact key are not seen."""
        if not self._planes_wanted():
            raise ValueError("compact_key: this engine's key switch reads raw key words (logN <= 12 or a checker backend)")
        if ksk.origin == types.origins["galk"]:          # a Galois key = one rotation key per power of two: each of them
            return sum(self.compact_key(k) for k in ksk.data)
        hit = self._key_packs.get(id(self._key_anchor(ksk)))
        if hit is None or hit["ref"]() is not self._key_anchor(ksk) or not hit.get("own"):
            raise ValueError("compact_key: not a key made by this engine (a foreign key's tensors are the caller's)")
        self._key_pack(ksk)                       # the planes copy, current
        freed = 0
        for pk in hit["packs"]:
            st = pk.untyped_storage()
            if st.nbytes():
                hit.setdefault("raw_bytes", {})[id(pk)] = st.nbytes()
                freed += st.nbytes()
                st.resize_(0)
        hit["compact"] = True
        return freed

    def expand_key(self, ksk):
        """Undo compact_key(): the raw pack is allocated again and filled from the planes copy — integer-class rows word for word,
        fp64-class rows with the CANONICAL residues of the words they held (a key's words are lazy Montgomery words, the planes keep
        their residues: the same key, every key switch the same words; byte-identical to the original only on integer-class rows)."""
        if ksk.origin == types.origins["galk"]:
            for k in ksk.data:
                self.expand_key(k)
            return
        hit = self._key_packs.get(id(self._key_anchor(ksk)))
        if hit is None or not hit.get("compact"):
            return
        loc = self._loc(0, special=True)
        for i, (d, pk) in enumerate(zip(loc, hit["packs"])):
            pk.untyped_storage().resize_(hit["raw_bytes"][id(pk)])
            planes = hit["planes"][i]
            q = self._consts(d, 0, True).q_host
            N = self.ctx.N
            for r in range(pk.size(2)):
                if int(q[r]) >= (1 << 41):
                    pk[:, :, r].copy_(planes[:, :, r])
                    continue
                lo = planes[:, 0, r].contiguous().view(torch.int32).view(-1, N // 2, 4).to(torch.int64) & 0xffffffff
                hi = planes[:, 1, r, :N // 2].contiguous().view(torch.int16).view(-1, N // 2, 4).to(torch.int64) & 0xffff
                w = (hi << 32) | lo                                            # [parts, N / 2, (b[j], b[j+1], a[j], a[j+1])]
                pk[:, 0, r].copy_(w[:, :, 0:2].reshape(-1, N))
                pk[:, 1, r].copy_(w[:, :, 2:4].reshape(-1, N))
        hit["compact"] = False
        hit["planes_ver"] = tuple(p_._version for p_ in hit["packs"])       # the copy_ above moved the counters: the planes are current

    def release_key(self, ksk):
        """Drop what the engine holds for a key-switch key beyond the key's own tensors, now: the packed copy of a foreign
        key; for a key made by this engine — whose tensors are views of its raw pack — the PLANES copy the fused key switch
        reads (a second tensor of the raw pack's size: gold ~ 430 MB per key, a Galois key set 15 x that).  Either is
        rebuilt on the key's next use and dropped when the key's tensors die."""
        hit = self._key_packs.get(id(self._key_anchor(ksk)))
        dropped = []
        if hit is not None and hit.get("compact"):
            self.expand_key(ksk)                 # the planes are the only copy of a compact key: the raw words come back first
        if hit is not None and hit.get("own"):
            dropped = hit.pop("planes", None) or []
            hit.pop("planes_ver", None)
        else:
            gone = self._key_packs.pop(id(self._key_anchor(ksk)), None)
            dropped = (gone or {}).get("packs", [])
        self._purge_graphs(self._tables, dropped)

    def invalidate_key(self, ksk):
        """Tell the engine that a key's words were changed behind torch's back (a write through a raw pointer does not move
        the version counters the engine watches): its planes copy / packed copy is rebuilt on the next use."""
        self.release_key(ksk)

    def _check_kernel_limits(self):
        """The fused kernels reserve registers for at most lf_limits() digit limbs / special primes / rows; a
        parameter set beyond them must fail HERE, not as a generic launch status deep inside a key switch."""
        lim = getattr(self.backend, "limits", None)
        if lim is None:
            return
        K = self.ntt.num_special_primes
        widest = max((len(part) for per_dev in self.ntt.p.p[0] for part in per_dev), default=0)
        rows = max((len(d) for d in self.ntt.p.destination_arrays_with_special[0]), default=0)
        if K > lim["special_primes"] or widest > lim["digit_limbs"]:
            raise errors.KernelLimitExceeded(
                f"num_special_primes = {K} (digits of up to {widest} limbs) exceeds what the key-switch kernels are "
                f"built for (at most {lim['special_primes']} special primes, {lim['digit_limbs']} limbs per digit)")
        if rows > lim["rows"]:
            raise errors.KernelLimitExceeded(f"{rows} limb rows on one device exceed the kernels' limit of {lim['rows']}")

    # =============================================================================================
    # encrypt / decrypt (eng.py:417-595, 1472-1688)
    # =============================================================================================
    def _encrypt_poly(self, encoded, pk, level, dc_rns=None):
        mult_type = -2 if pk.include_special else -1
        e0e1 = self._live(self.rng.discrete_gaussian(repeats=2))
        e0 = self.ntt.tile_unsigned([e[0] for e in e0e1], level, mult_type)
        e1 = self.ntt.tile_unsigned([e[1] for e in e0e1], level, mult_type)
        pt = self.ntt.tile_unsigned(encoded, level, mult_type)
        if dc_rns is not None:
            for t, dc in zip(pt, dc_rns):
                t[:, 0] += dc
        self.ntt.mont_enter_scale(pt, level, mult_type)
        self.ntt.mont_redc(pt, level, mult_type)
        pte0 = self.ntt.mont_add(pt, e0, level, mult_type)

        loc = self._loc(level, special=pk.include_special)
        start = self.ntt.starts[level]
        pk0 = [pk.data[0][i][start[d]:] for i, d in enumerate(loc)]
        pk1 = [pk.data[1][i][start[d]:] for i, d in enumerate(loc)]
        v = self._live(self.rng.randint(amax=2, shift=0, repeats=1))
        v = self.ntt.tile_unsigned(v, level, mult_type)
        self.ntt.enter_ntt(v, level, mult_type)
        vpk0 = self.ntt.mont_mult(v, pk0, level, mult_type)
        vpk1 = self.ntt.mont_mult(v, pk1, level, mult_type)
        self.ntt.intt_exit(vpk0, level, mult_type)
        self.ntt.intt_exit(vpk1, level, mult_type)
        ct0 = self.ntt.mont_add(vpk0, pte0, level, mult_type)
        ct1 = self.ntt.mont_add(vpk1, e1, level, mult_type)
        self.ntt.reduce_2q(ct0, level, mult_type)
        self.ntt.reduce_2q(ct1, level, mult_type)
        return self._new((ct0, ct1), types.origins["ct"], level=level, include_special=mult_type == -2)

    @errors.log_error
    def encrypt(self, pt: list[torch.Tensor], pk: data_struct, level: int = 0) -> data_struct:
        if pk.origin != types.origins["pk"]:
            raise errors.NotMatchType(origin=pk.origin, to=types.origins["pk"])
        return self._encrypt_poly([p.clone() for p in pt], pk, level)

    def encodecrypt(self, m, pk: data_struct, level: int = 0, padding=True) -> data_struct:
        if pk.origin != types.origins["pk"]:
            raise errors.NotMatchType(origin=pk.origin, to=types.origins["pk"])
        if padding:
            m = self.padding(m=m)
        dev0 = self.ntt.devices[self.local_ids[0]]
        pt = encdec.encode(m, scale=self.scale, device=dev0, norm=self.norm, deviation=self.deviations[level],
                           rng=self.rng, return_without_scaling=self.bias_guard)
        dc_rns = None
        if self.bias_guard:
            dc_integral = pt[0].item() // 1
            pt[0] -= dc_integral
            dc_scale = int(dc_integral) * int(self.scale)
            dc_rns = [self._t64([dc_scale % self.ctx.q[i] for i in self.ntt.p.destination_arrays[level][d]], d)
                      for d in self._loc(level)]
            pt = self.rng.randround(pt * np.float64(self.scale))
        return self._encrypt_poly(self._replicate_plain(pt), pk, level, dc_rns)

    def _decrypt_rows(self, ct, sk, li=0, dev=0):
        """pt rows (c0 + c1*s, or the triplet form), canonical, of the local shard `li` = logical device `dev`: device 0 as in the
        reference (eng.py:481-566); the balanced limb map also asks for device 1's (see _scaler_dev)."""
        level = ct.level
        if li or dev:
            return _OneShard(self, li, dev)._decrypt_rows(ct, sk)
        return self._decrypt_rows_on(self.ntt, ct, sk, 0, 0)

    @staticmethod
    def _decrypt_rows_on(n, ct, sk, li, dev):
        level = ct.level
        sk0 = sk.data[li][n.starts[level][dev]:]
        if ct.origin == types.origins["ct"]:
            if ct.ntt_state or ct.montgomery_state:
                raise errors.NotMatchDataStructState(origin=ct.origin)
            a = ct.data[1][li].clone()
            n.enter_ntt([a], level)
            sa = n.mont_mult([a], [sk0], level)
            n.intt_exit(sa, level)
            pt = n.mont_add([ct.data[0][li]], sa, level)
        elif ct.origin == types.origins["ctt"]:
            if not ct.ntt_state or not ct.montgomery_state:
                raise errors.NotMatchDataStructState(origin=ct.origin)
            d0 = [ct.data[0][li].clone()]
            n.intt_exit_reduce(d0, level)
            d1_s = n.mont_mult([ct.data[1][li]], [sk0], level)
            s2 = n.mont_mult([sk0], [sk0], level)
            d2_s2 = n.mont_mult([ct.data[2][li]], s2, level)
            n.intt_exit(d1_s, level)
            n.intt_exit(d2_s2, level)
            pt = n.mont_add(d0, d1_s, level)
            pt = n.mont_add(pt, d2_s2, level)
        else:
            raise errors.NotMatchType(origin=ct.origin, to=f"{types.origins['ct']} or {types.origins['ctt']}")
        n.reduce_2q(pt, level)
        return pt

    def _need_final_scalar(self, level):
        if self.final_scalar[level] is None:
            raise ValueError(f"decryption at level {level}: the base prime is the only limb left, there is no scale prime to scale by")

    def _final_scale(self, base, scaler, level, final_round):
        """(base - scaler) * q_l^-1 mod base prime, centred, + rounding bit (eng.py:517-533)."""
        self._need_final_scalar(level)
        scaled = self.ntt.mont_sub([base], [scaler], -1)
        self.ntt.mont_enter_scalar(scaled, [self.final_scalar[level]], -1)
        self.ntt.reduce_2q(scaled, -1)
        self.ntt.make_signed(scaled, -1)
        if final_round:
            # (the reference compares against device 0's last scale prime whatever the scaler's own prime is — kept, for parity;
            # with the scaler fetched from device 1 — balanced limb map — it is the scaler's prime)
            rounding_prime = self.final_q[level] if self._scaler_dev[level] else self.ntt.qlists[0][-self.ctx.num_special_primes - 2]
            scaled[0] += (scaler[0] > (rounding_prime // 2)) * 1
        return scaled

    def decrypt(self, ct: data_struct, sk: data_struct, final_round=True) -> list[torch.Tensor]:
        if sk.origin != types.origins["sk"]:
            raise errors.NotMatchType(origin=sk.origin, to=types.origins["sk"])
        if not sk.ntt_state or not sk.montgomery_state:
            raise errors.NotMatchDataStructState(origin=sk.origin)
        got = self._base_and_scaler(ct, sk)
        if got is None:
            return None  # the base-prime row lives on device 0
        base, scaler, _ = got
        return self._final_scale(base, scaler, ct.level, final_round)

    def _base_and_scaler(self, ct, sk):
        """(base-prime row, scaler row, pt rows of device 0) on device 0, or None on a rank that does not hold device 0.  The
        scaler is device 0's first row as in the reference — or, in the balanced limb map once device 0 has lost its scale
        digit, the first row of device 1 (_scaler_dev): computed there and moved (one process: a device copy; one process per
        GPU: one message from rank 1 to rank 0; ranks 0 and 1 both call decrypt, every other rank returns at once)."""
        level = ct.level
        src = self._scaler_dev[level]
        base_at = -self.ctx.num_special_primes - 1 if ct.include_special else -1
        if src == 0:
            if 0 not in self.local_ids:
                return None
            pt = self._decrypt_rows(ct, sk)
            return pt[0][base_at][None, :], pt[0][0][None, :], pt
        N = self.ctx.N
        if self._multi:
            me = self.local_ids[0]
            if me not in (0, src):
                return None
            buf = self._ws("scaler_row", (1, N), me)
            pt = self._decrypt_rows(ct, sk, 0, me)
            if me == src:
                buf[0].copy_(pt[0][0])
            self.comm.fanout_into(buf, src, [0, src])
            if me != 0:
                return None
            return pt[0][base_at][None, :], buf, pt
        pt = self._decrypt_rows(ct, sk)
        other = self._decrypt_rows(ct, sk, src, src)
        return pt[0][base_at][None, :], other[0][0][None, :].to(pt[0].device), pt

    def decrypt_double(self, ct, sk, final_round=True):
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        return self.decrypt(ct, sk, final_round)

    def decrypt_triplet(self, ct_mult, sk, final_round=True):
        if ct_mult.origin != types.origins["ctt"]:
            raise errors.NotMatchType(origin=ct_mult.origin, to=types.origins["ctt"])
        return self.decrypt(ct_mult, sk, final_round)

    def decryptcode(self, ct: data_struct, sk: data_struct, is_real=False, final_round=True):
        if (not sk.ntt_state) or (not sk.montgomery_state):
            raise errors.NotMatchDataStructState(origin=sk.origin)
        level = ct.level
        got = self._base_and_scaler(ct, sk)
        if got is None:
            return None
        base, scaler, pt = got
        base_at = -self.ctx.num_special_primes - 1 if ct.include_special else -1
        dest0 = self.ntt.p.destination_arrays[level][0]
        guard = len(dest0) >= 3 and self.bias_guard and self._scaler_dev[level] == 0
        if guard:
            # the DC coefficient is rebuilt from three residues by CRT in Python ints (eng.py:1616-1646)
            dc0, dc1, dc2 = base[0][0].item(), scaler[0][0].item(), pt[0][1][0].item()
            base[0][0] = 0
            scaler[0][0] = 0
            q0, q1, q2 = (self.ctx.q[dest0[base_at]], self.ctx.q[dest0[0]], self.ctx.q[dest0[1]])
            Q, Q0, Q1, Q2 = q0 * q1 * q2, q1 * q2, q0 * q2, q0 * q1
            dc = (dc0 * pow(Q0, -1, q0) * Q0 + dc1 * pow(Q1, -1, q1) * Q1 + dc2 * pow(Q2, -1, q2) * Q2) % Q
            dc = dc if dc <= Q // 2 else dc - Q
            dc = (dc + (q1 - 1)) // q1
        scaled = self._final_scale(base, scaler, level, final_round)
        correction = self.corrections[level]
        decoded = encdec.decode(scaled[0][-1], scale=self.scale, correction=correction, norm=self.norm,
                                return_without_scaling=self.bias_guard)
        decoded = decoded[:self.ctx.N // 2].cpu().numpy()
        decoded = decoded / self.scale * correction
        if guard:
            decoded += dc / self.scale * correction
        return decoded.real if is_real else decoded

    def encorypt(self, m, pk: data_struct, level: int = 0, padding=True):
        return self.encodecrypt(m, pk=pk, level=level, padding=padding)

    def decrode(self, ct: data_struct, sk: data_struct, is_real=False, final_round=True):
        return self.decryptcode(ct=ct, sk=sk, is_real=is_real, final_round=final_round)

    # =============================================================================================
    # exchange steps
    # =============================================================================================
    def _share_rows(self, rows_by_owner, owner, targets, shape):
        """Row block held by `owner` -> {device: tensor} on every target device of THIS process (one process driving
        several devices: device-to-device copies where the reference stages through pinned host memory,
        eng.py:999-1011).  With one process per GPU the rows travel by an in-place RCCL broadcast instead
        (_rescale_operands)."""
        out = {}
        for d in targets:
            dev = self.ntt.devices[d]
            out[d] = rows_by_owner if str(rows_by_owner.device) == dev else rows_by_owner.to(dev)
        return out

    # =============================================================================================
    # rescale (eng.py:967-1052)
    # =============================================================================================
    def _rescale_into(self, ct, outs, exact_rounding=True):
        """Rescale ct (level l) writing component c of local device d into outs[c][d] ([rows, N] views)."""
        self._rescale_many([ct], [outs], exact_rounding)

    def _rescale_operands(self, cts, exact_rounding=True):
        """What the rescale of several ciphertexts of one level reads, per local device of the next level:
        {d: (sources, dropped-limb rows)} in the order (ct 0 comp 0, ct 0 comp 1, ct 1 comp 0, ..), and the
        rounding threshold.  Without exact rounding the [row0 > q_l / 2] term is dropped (eng.py:1017-1027,
        1036-1038): the kernels' threshold is put out of reach."""
        level = cts[0].level
        nxt = level + 1
        owner = self.ntt.p.rescaler_loc[level]
        loc_before = self._loc(level)
        N = self.ctx.N
        targets = list(range(self.len_devices[nxt]))
        round_at = self.ctx.q[self.ntt.p.destination_arrays[level][owner][0]] // 2 if exact_rounding else (1 << 62)
        # the dropped limb's row of every polynomial, on every local target device
        rows0 = []                                   # rows0[k][comp] = {device: [N] tensor}
        multi = self._multi
        if multi:
            # the dropped limb's rows of ALL operands (cc_mult: both ciphertexts, both components) in ONE message, in
            # place on a buffer kept per operand count
            me = self.local_ids[0]
            buf = self._ws("rescale_rows", (2 * len(cts), N), me)
            if owner == me:
                i = loc_before.index(owner)
                rows = [ct.data[comp][i][0] for ct in cts for comp in range(2)]
                if hasattr(self.backend, "gather_rows") and len(rows) <= 8 and all(r.is_contiguous() and r.data_ptr() % 16 == 0 for r in rows):
                    self.backend.gather_rows(rows, buf)      # one launch instead of a copy per row
                else:
                    for k, r in enumerate(rows):
                        buf[k].copy_(r)
            self.comm.fanout_into(buf, owner, targets)   # one message per rank that still holds rows at the next level
            for k in range(len(cts)):
                rows0.append([{d: buf[2 * k + comp] for d in targets if d in self.local_ids} for comp in range(2)])
        else:
            i = loc_before.index(owner)
            for ct in cts:
                rows0.append([self._share_rows(ct.data[comp][i][0], owner, targets, (N,)) for comp in range(2)])
        per_dev = {}
        for d in self._loc(nxt):
            i = loc_before.index(d)
            srcs, r0s = [], []
            for k, ct in enumerate(cts):
                for comp in range(2):
                    src = ct.data[comp][i]
                    srcs.append(src[1:] if d == owner else src)
                    r0s.append(rows0[k][comp][d])
            per_dev[d] = (srcs, r0s)
        return per_dev, round_at

    def _rescale_many(self, cts, outs_list, exact_rounding=True):
        """Rescale several ciphertexts of one level; outs_list[k][c][d] receives component c of cts[k] on local
        device d.  All their polynomials go through ONE launch per device."""
        level = cts[0].level
        nxt = level + 1
        per_dev, round_at = self._rescale_operands(cts, exact_rounding)
        for d, (srcs, r0s) in per_dev.items():
            dsts = [outs_list[k][comp][d] for k in range(len(cts)) for comp in range(2)]
            self.backend.rescale_batch(srcs, r0s, dsts, self._rows(d, nxt, False), self.rescale_scales[level][d], round_at,
                                       self._consts(d, nxt, False))

    def rescale(self, ct: data_struct, exact_rounding=True) -> data_struct:
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        nxt = ct.level + 1
        if nxt >= self.num_levels:
            raise errors.MaximumLevelError(level=ct.level, level_max=self.num_levels)
        loc = self._loc(nxt)
        outs = [{d: torch.empty((self._rows(d, nxt, False), self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
                 for d in loc} for _ in range(2)]
        self._rescale_into(ct, outs, exact_rounding)
        return self._new(([outs[0][d] for d in loc], [outs[1][d] for d in loc]), types.origins["ct"], level=nxt)

    # =============================================================================================
    # whole ops behind one native call (include/ckks_hip.h: lf_ks_plan, lf_cc_mult_evk, lf_switch_key)
    # =============================================================================================
    def _native_level(self, level):
        """The device index if every limb of `level` lives on ONE device of this process and the backend has the one-call
        entries; None otherwise (several devices / ranks: the exchange steps sit between the launches)."""
        if not getattr(self.backend, "native_ops", False) or self.ctx.logN < self.backend.fused_ks_min_logN:
            return None
        if self.len_devices[level] != 1 or (self._multi and self.comm.world_size == 1):   # (solo_sharded: the halves, not the one call)
            return None
        loc = self._loc(level)
        return loc[0] if len(loc) == 1 else None

    def _op_plan(self, level, d, nct=1):
        """lf_ks_plan of (level, device): constants, tables and scratch of a key switch AT `level` and of a cc_mult INTO it,
        resolved once (per pipeline lane: the scratch tensors are per lane).  nct = ciphertexts per batched call the scratch
        is sized for (1: the single-ciphertext workspaces the step-by-step path uses too).  `d` may be this rank's device of a
        limb-sharded engine: the plan then describes its rows, `nparts` all digits, `dig_nparts` the digits it owns."""
        self._same_stream(d)
        key = ("plan", level, d, self._lane, nct)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        N, K = self.ctx.N, self.ntt.num_special_primes
        rows, ell = self._rows(d, level, True), self._rows(d, level, False)
        tabs = self._ks_tables(level)
        cs = self._consts(d, level, True)
        nparts = len(tabs["order"])
        dig_nparts, dig_desc, dig_tab = tabs[("digits", d)]
        desc, E, Ed = tabs[("extend", d)]
        words = self.backend.moddown_ws_words(2 * nct, ell, K, N)
        ints = {"logN": self.ctx.logN, "ell": ell, "K": K, "nparts": nparts, "dig_nparts": dig_nparts, "md_ws_words": words,
                "round_at": 0, "max_nct": nct}
        if nct == 1:
            scratch = {"state": self._ws("ks_state", (ell, N), d), "ext": self._ws("ks_ext", (nparts, rows, N), d),
                       "sum": self._ws("ks_sum", (2, rows, N), d), "md_ws": self._ws("ks_moddown_plan", (words,), d),
                       "x4": self._ws("mult4", (4, ell, N), d), "d2": self._ws("mult_d2", (ell, N), d)}
        else:
            scratch = {"state": self._ws("ks_state_batch", (nct, ell, N), d), "ext": self._ws("ks_ext_batch", (nct, nparts, rows, N), d),
                       "sum": self._ws("ks_sum_batch", (nct, 2, rows, N), d), "md_ws": self._ws("ks_moddown_plan", (words,), d),
                       "x4": self._ws("multx", (nct, 4, ell, N), d), "d2": self._ws("multx_d2", (nct, ell, N), d)}
        tensors = {"Rs": self._vec("Rs", d, level, True), "Ninv": self._vec("Ninv", d, level, True), "dig_desc": dig_desc,
                   "dig_tab": dig_tab, "ext_desc": desc, "E": E, "Ed": Ed, "PiR": tabs[("pir", d)], "PiP": tabs[("pip", d)],
                   "own": tabs[("own", d)], "rescale_scales": None, "PR": self._PR(d, level), **scratch}
        if level >= 1:
            owner = self.ntt.p.rescaler_loc[level - 1]
            ints["round_at"] = self.ctx.q[self.ntt.p.destination_arrays[level - 1][owner][0]] // 2
            tensors["rescale_scales"] = self.rescale_scales[level - 1][d]
        plan, keep = self.backend.make_plan(ints, tensors, cs.q_host, self._tw(d, level, True), self._tw(d, level, True, True), cs)
        hit = self._tables[key] = (plan, keep, tabs["first_part"], self.ntt.starts[level][d])
        return hit

    def _cc_mult_native(self, a, b, evk, level, d):
        """cc_mult + relinearize as ONE native call (lf_cc_mult_evk); None if an operand is not laid out for it."""
        N = self.ctx.N
        ins, row0s = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()
        k = 0
        for ct in (a, b):
            for comp in range(2):
                t = ct.data[comp][0]
                if not t.is_contiguous() or t.dtype != torch.int64:
                    return None
                ptr = t.data_ptr()
                row0s[k], ins[k] = ptr, ptr + N * 8      # the dropped limb is the first row; the survivors follow it
                k += 1
        plan, _, first_part, row_off = self._op_plan(level, d)
        kpack = self._key_pack(evk)[self._loc(0, special=True).index(d)]
        out = torch.empty((2, plan.ell, N), dtype=torch.int64, device=self.ntt.devices[d])
        self.backend.cc_mult_evk(plan, ins, row0s, kpack, first_part, row_off, out)
        return self._new(([out[0]], [out[1]]), types.origins["ct"], level=level)

    def _sharded_native(self, level):
        """This rank's device if the level is limb-sharded over ranks (one process per GPU), this rank holds rows of it and
        the backend has the native halves of an op (lf_*_pre / lf_ks_plan_fwd / lf_*_post); None otherwise."""
        if not self._multi or not getattr(self.backend, "native_ops", False):
            return None
        if self.ctx.logN < self.backend.fused_ks_min_logN or not hasattr(self.backend, "cc_mult_pre"):
            return None
        loc = self._loc(level)
        return loc[0] if len(loc) == 1 and (self.len_devices[level] > 1 or self.comm.world_size == 1) else None

    def _sharded_schedule(self, level, d):
        """The digit exchange of this rank at `level` as plain data: (digit buffer in storage order, pieces and peers of
        comm.exchange_rows, [(buffer row, rows, state row)] of the digits this rank owns, its own runs [(first digit, digits)],
        the foreign runs)."""
        key = ("sched", level, d)
        hit = self._tables.get(key)
        if hit is None:
            tabs = self._ks_tables(level)
            groups, nparts = tabs["groups"], len(tabs["order"])
            own_rows = [(row0, nrows, src_row) for owner, first, count, row0, nrows, src_row in groups if owner == d]
            own_runs = [(first, count) for owner, first, count, _, _, _ in groups if owner == d]
            foreign, run = [], None
            for owner, first, count, _, _, _ in groups + [(d, nparts, 0, 0, 0, 0)]:
                if owner == d:
                    if run is not None:
                        foreign.append(run)
                        run = None
                else:
                    run = (first, count) if run is None else (run[0], run[1] + count)
            hit = self._tables[key] = ([(g[0], g[3], g[4]) for g in groups], list(range(self.len_devices[level])), own_rows, own_runs,
                                       foreign, tabs["total_rows"])
        return hit

    def _sharded_forward(self, plan, d, level, relin):
        """The digit exchange between the halves of a sharded op: this rank's digits (plan.state) go out as one batch of
        point-to-point messages while it extends + transforms the digits it owns; the foreign runs follow the single wait."""
        tabs = self._ks_tables(level)
        state = self._ws("ks_state", (plan.ell, self.ctx.N), d)          # = plan.state
        dig, ready = self._exchange_digits({d: state}, level, tabs)[d]
        for handle, first, count in ready:
            if handle is not None:
                handle.wait()
            self.backend.plan_fwd(plan, dig, first, count, relin)

    @staticmethod
    def _capture(fn, device):
        """fn() — launches on torch's current stream, fixed addresses only — as a HIP graph: one warm-up run on a side stream
        (lazy tables), then the capture.  Returns the graph (replay() enqueues it on the current stream)."""
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make meanwhile (a communicator's proxy thread) do not invalidate the capture
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            fn()
        return g

    def _sharded_segments(self, kind, level, d, plan, kpack, first_part, row_off, capture=False):
        """The launches of a rank's half-ops that only touch the plan's scratch, the tables and the key, captured ONCE per
        (op kind, level, lane, key pack) into three HIP graphs around the digit exchange:
            a1  cc_mult: the rest of `pre` (tiled pass of the operands, x1 * y1, inverse NTT, digits); both kinds: this rank's
                digit rows into the storage-order exchange buffer;
            a2  extension + forward NTT of the digits it owns (runs beside the exchange);
            c   the same for the foreign runs, then inner product + inverse NTT.
        What stays eager is what reads or writes caller tensors: the first launch (operands) and the mod-down (result), and
        the exchange itself.  A replay costs 6-8 us of host time whatever it holds (tools/graph_host_cost.py) against 4.4 us per
        launch: a rank of a gold cc_mult over 8 GPUs enqueues in ~55 us instead of 140 (profiles/r05_host_overhead.txt).
        None when the device is not a HIP device (the CPU checker backend), and — unless `capture` — when the segments of this
        (kind, level, lane, key) have not been captured yet: the FIRST op of each runs eagerly and captures when it is
        complete (capture=True, after its mod-down) — the warm-up run then works on the valid scratch that op left, and the
        device-wide synchronisation of a capture never falls between the two exchanges of an op its peers are inside."""
        dev = self.ntt.devices[d]
        if not self.graph_sharded or not str(dev).startswith("cuda") or not hasattr(torch.cuda, "CUDAGraph"):
            return None
        fmt = getattr(kpack, "lf_key_format", 0)
        key = ("sgraph", kind, level, d, self._lane, kpack.data_ptr(), tuple(kpack.stride()), fmt, first_part, row_off)
        hit = self._tables.get(key)
        if hit is not None or not capture:
            return hit
        relin = kind == "mult"
        pieces, peers, own_rows, own_runs, foreign, total_rows = self._sharded_schedule(level, d)
        N = self.ctx.N
        state = self._ws("ks_state", (plan.ell, N), d)
        buf = self._ws("ks_digits_all", (total_rows, N), d)
        from .backend import _ds
        be = self.backend

        def a1():
            if relin:
                be.cc_mult_pre(plan, None, None, _ds(buf)[1], which=2)
            for row0, nrows, src_row in own_rows:
                buf[row0:row0 + nrows].copy_(state[src_row:src_row + nrows])

        def a2():
            for first, count in own_runs:
                be.plan_fwd(plan, buf, first, count, relin)

        def c():
            for first, count in foreign:
                be.plan_fwd(plan, buf, first, count, relin)
            if relin:
                be.cc_mult_post(plan, kpack, first_part, row_off, None, which=1)
            else:
                be.switch_key_post(plan, None, 0, False, kpack, first_part, row_off, None, which=1)

        try:
            hit = {"a1": self._capture(a1, dev) if (relin or own_rows) else None,
                   "a2": self._capture(a2, dev) if own_runs else None, "c": self._capture(c, dev),
                   "buf": buf, "pieces": pieces, "peers": peers, "keep": (state, kpack)}
        except Exception as e:     # a capture that fails costs nothing but the replays: this rank goes on with eager launches
            import warnings        # (the exchanges it issues are the same either way, so its peers are not affected)
            warnings.warn(f"HIP-graph capture of the sharded op segments failed ({type(e).__name__}: {e}); eager launches from here on")
            self.graph_sharded = False
            torch.cuda.synchronize(dev)
            return None
        self._tables[key] = hit
        return hit

    def _sharded_exchange_replay(self, seg):
        """a1 -> [exchange, asynchronous] -> a2 beside it -> wait -> c."""
        if seg["a1"] is not None:
            seg["a1"].replay()
        handle = self.comm.exchange_rows(seg["buf"], seg["pieces"], seg["peers"])
        if seg["a2"] is not None:
            seg["a2"].replay()
        handle.wait()
        seg["c"].replay()

    def _cc_mult_sharded_native(self, a, b, evk, level, d):
        """cc_mult + relinearize of a limb-sharded level with the host side of this rank in three native calls + one per run
        of digits (lf_cc_mult_evk_pre, lf_ks_plan_fwd, lf_cc_mult_evk_post) around the two exchanges — on a HIP device the
        fixed-address launches among them replayed from three HIP graphs (_sharded_segments).  Always completes: every
        rank issues exactly one rescale exchange and one digit exchange per op, whatever the layout of its operands."""
        N = self.ctx.N
        # exchange 1: the dropped limb's rows of the four polynomials, one message from their owner (the lean form of
        # _rescale_operands for ONE local device: pointers only, the per-level facts cached)
        lvl = a.level
        facts = self._tables.get(("rs_sharded", lvl, d))
        if facts is None:
            owner = self.ntt.p.rescaler_loc[lvl]
            facts = self._tables[("rs_sharded", lvl, d)] = (owner, self._loc(lvl).index(d), list(range(self.len_devices[level])),
                                                           self.ctx.q[self.ntt.p.destination_arrays[lvl][owner][0]] // 2)
        owner, i, targets, round_at = facts
        # never bail out from here: the exchange happens on every rank, and contiguity is a per-rank property — a rank that
        # fell back to the step-by-step path would repeat it while its peers go on to the digit exchange
        polys = [t if t.is_contiguous() else t.contiguous() for t in (a.data[0][i], a.data[1][i], b.data[0][i], b.data[1][i])]
        rbuf = self._ws("rescale_rows", (4, N), d)
        if owner == d:
            if all(t.data_ptr() % 16 == 0 for t in polys):
                self.backend.gather_rows([t[0] for t in polys], rbuf)      # one launch instead of a copy per row
            else:
                for k, t in enumerate(polys):
                    rbuf[k].copy_(t[0])
        self.comm.fanout_into(rbuf, owner, targets)
        plan, _, first_part, row_off = self._op_plan(level, d)
        assert plan.round_at == round_at
        skip = N * 8 if owner == d else 0                                  # the owner's surviving rows start behind the dropped one
        ins = (ctypes.c_void_p * 4)(*[t.data_ptr() + skip for t in polys])
        r0 = rbuf.data_ptr()
        row0s = (ctypes.c_void_p * 4)(r0, r0 + N * 8, r0 + 2 * N * 8, r0 + 3 * N * 8)
        srcs = polys
        from .backend import _ds
        kpack = self._key_pack(evk)[self._loc(0, special=True).index(d)]
        out = torch.empty((2, plan.ell, N), dtype=torch.int64, device=self.ntt.devices[d])
        seg = self._sharded_segments("mult", level, d, plan, kpack, first_part, row_off)
        if seg is not None:
            self.backend.cc_mult_pre(plan, ins, row0s, _ds(srcs[0])[1], which=1)     # eager: reads the operands
            self._sharded_exchange_replay(seg)                                      # exchange 2: the digits
            self.backend.cc_mult_post(plan, kpack, first_part, row_off, out, which=2)   # eager: the mod-down writes the result
            return self._new(([out[0]], [out[1]]), types.origins["ct"], level=level)
        self.backend.cc_mult_pre(plan, ins, row0s, _ds(srcs[0])[1])
        self._sharded_forward(plan, d, level, True)                      # exchange 2: the digits
        self.backend.cc_mult_post(plan, kpack, first_part, row_off, out)
        self._sharded_segments("mult", level, d, plan, kpack, first_part, row_off, capture=True)   # the next op replays
        return self._new(([out[0]], [out[1]]), types.origins["ct"], level=level)

    # =============================================================================================
    # multiplication (eng.py:1072-1151)
    # =============================================================================================
    def cc_mult(self, a: data_struct, b: data_struct, evk: data_struct, relin=True) -> data_struct:
        if a.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=a.origin, to=types.origins["sk"])
        if b.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=b.origin, to=types.origins["sk"])
        level = a.level + 1
        if level >= self.num_levels:
            raise errors.MaximumLevelError(level=a.level, level_max=self.num_levels)
        loc = self._loc(level)
        N, logN = self.ctx.N, self.ctx.logN
        if relin and a.level == b.level and not (a.ntt_state or b.ntt_state or a.include_special or b.include_special):
            d = self._native_level(level)
            if d is not None and self._native_level(a.level) == d:
                out = self._cc_mult_native(a, b, evk, level, d)
                if out is not None:
                    return out
            d = self._sharded_native(level)
            if d is not None and getattr(self.backend, "relin_fold", False):
                out = self._cc_mult_sharded_native(a, b, evk, level, d)
                if out is not None:
                    return out
        d0, d1, d2 = [], [], []
        stacks = {d: self._ws("mult4", (4, self._rows(d, level, False), N), d) for d in loc}
        # x0, x1, y0, y1: both rescales and the four forward transforms are one backend call per device (for
        # two-pass ring degrees the rescale is evaluated inside the first NTT pass).  With relinearisation the
        # triplet never leaves this method: only residues matter, so the 40-bit limbs take the relaxed
        # plain-domain transforms (one fp64 product per tensor term)
        per_dev, round_at = self._rescale_operands([a, b])
        fold = relin and logN >= self.backend.fused_ks_min_logN and getattr(self.backend, "relin_fold", False)
        for d in loc:
            rows = self._rows(d, level, False)
            c = self._consts(d, level, False)
            x = stacks[d]
            srcs, r0s = per_dev[d]
            self.backend.rescale_ntt(srcs, r0s, x, rows, logN, self.rescale_scales[a.level][d], round_at,
                                     self._tw(d, level, False), self._vec("Rs", d, level, False), c,
                                     relaxed=relin, plain=relin)
            if fold:
                # d2 = x1 * y1 goes straight into its inverse transform (the product is formed as the first pass reads
                # its tiles); d0 and d1 are never formed: the key switch adds P * d0 and P * d1 to its sums in the NTT
                # domain (dividing by P is linear and P * d vanishes modulo the special primes), so the triplet costs ONE
                # inverse transform instead of three, no tensor launch and no addend pass
                t = self._ws("mult_d2", (rows, N), d)
                self.backend.intt_mul(t, x[1], x[3], 1, rows, logN, self._tw(d, level, False, True),
                                      self._vec("Ninv", d, level, False), c)
                d2.append(t)
                continue
            out = torch.empty((3, rows, N), dtype=torch.int64, device=self.ntt.devices[d])
            self.backend.tensor(x[0], x[1], x[2], x[3], out[0], out[1], out[2], rows, c, plain=relin)
            d0.append(out[0]); d1.append(out[1]); d2.append(out[2])
        if fold:
            tabs = self._ks_tables(level)
            c0, c1 = self.create_switcher(d2, evk, level, fold={d: (stacks[d], self._PR(d, level), tabs[("own", d)]) for d in loc})
            return self._new((c0, c1), types.origins["ct"], level=level)
        ct_mult = self._new((d0, d1, d2), types.origins["ctt"], level=level, ntt_state=True, montgomery_state=True)
        return self._relinearize(ct_mult, evk, plain=True) if relin else ct_mult

    def _PR(self, dev, level):
        """P * R mod q over device `dev`'s ordinary rows at `level` (make_mont_PR, eng.py:252-263)."""
        key = ("PR", dev, level)
        t = self._tables.get(key)
        if t is None:
            t = self._tables[key] = self.mont_PR[dev][self.ntt.starts[level][dev]:]
        return t

    def relinearize(self, ct_triplet: data_struct, evk: data_struct) -> data_struct:
        if ct_triplet.origin != types.origins["ctt"]:
            raise errors.NotMatchType(origin=ct_triplet.origin, to=types.origins["ctt"])
        if not ct_triplet.ntt_state or not ct_triplet.montgomery_state:
            raise errors.NotMatchDataStructState(origin=ct_triplet.origin)
        return self._relinearize(ct_triplet, evk, plain=False)

    def _relinearize(self, ct_triplet, evk, plain):
        """plain: the triplet comes from cc_mult's internal plain-domain product (40-bit limbs un-Montgomeried)."""
        d0, d1, d2 = ct_triplet.data
        level = ct_triplet.level
        # the three inverse transforms mutate the triplet in place, as the reference does (eng.py:1127-1129)
        for i, d in enumerate(self._loc(level)):
            rows, c = self._rows(d, level, False), self._consts(d, level, False)
            tw, ninv = self._tw(d, level, False, True), self._vec("Ninv", d, level, False)
            stacked = self._as_stack([d0[i], d1[i], d2[i]])
            if stacked is not None:
                self.backend.intt(stacked, 3, rows, self.ctx.logN, tw, ninv, 2, c, relaxed=plain, plain=plain)
            else:
                for t in (d0[i], d1[i], d2[i]):
                    self.backend.intt(t, 1, rows, self.ctx.logN, tw, ninv, 2, c, relaxed=plain, plain=plain)
        c0, c1 = self.create_switcher(d2, evk, level, addends=(d0, d1))
        return self._new((c0, c1), types.origins["ct"], level=level)

    @staticmethod
    def _as_stack(ts):
        """The [k, rows, N] tensor whose slices are `ts`, if they are laid out back to back."""
        t0 = ts[0]
        step = t0.numel() * t0.element_size()
        if all(t.is_contiguous() and t.shape == t0.shape and t.data_ptr() == t0.data_ptr() + i * step
               and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for i, t in enumerate(ts)):
            return torch.as_strided(t0, (len(ts),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()))
        return None

    # =============================================================================================
    # hybrid key switching (eng.py:654-961)
    # =============================================================================================
    def _ks_tables(self, level):
        """Per-level descriptor tables of the fused key-switch kernels (built once, cached)."""
        key = ("ks", level)
        if key in self._tables:
            return self._tables[key]
        p, ctx = self.ntt.p, self.ctx
        n_alive = self.len_devices[level]
        # digits in storage order: (owner device, level-local rows, prime indices)
        digits = {}
        for d in range(n_alive):
            for i, rows in enumerate(p.parts[level][d][:-1]):
                digits[self.stor_ids[level][d][i]] = (d, rows, p.destination_parts[level][d][i])
        order = [digits[s] for s in range(len(digits))]
        row_start, acc = [], 0
        for _, rows, _ in order:
            row_start.append(acc)
            acc += len(rows)
        tabs = {"order": order, "row_start": row_start, "total_rows": acc, "first_part": min(
            min(a) for a in self.parts_alloc[level] if len(a) > 0)}

        for d in self._loc(level):
            # (1) Garner constants of this device's own digits
            desc, flat = [], []
            for i, rows in enumerate(p.parts[level][d][:-1]):
                Y, Ls, _ = self.ntt.digit_constants(p.destination_parts[level][d][i])
                y_off = len(flat); flat += Y
                l_off = len(flat); flat += [x for row in Ls for x in row]
                desc.append([rows[0], len(rows), y_off, l_off])
            tabs[("digits", d)] = (len(desc), self._t64(desc, d), self._t64(flat if flat else [0], d))
            # (2) extension constants of EVERY digit onto this device's rows
            dest = p.destination_arrays_with_special[level][d]
            desc, flat = [], []
            for s, (_, rows, primes) in enumerate(order):
                m = [ctx.q[i] for i in primes]
                e_off = len(flat)
                L = 1
                for i in range(len(m)):
                    flat += [L * ctx.R_square[r] % ctx.q[r] for r in dest]
                    L *= m[i]
                wide = any(ctx.q[i] >= (1 << 41) for i in primes)   # digit words beyond fp64's 53 bits
                desc.append([row_start[s], len(rows) | (int(wide) << 8), e_off])
            # the same constants as plain residues in doubles (fp64 class of the fused core): L_{i-1} mod q_r
            plain = []
            for s_, (_, rows, primes) in enumerate(order):
                L = 1
                for i in range(len(primes)):
                    plain += [float(L % ctx.q[r]) for r in dest]
                    L *= ctx.q[primes[i]]
            # behind them, at the same offsets, the digit's own primes m_i mod q_r: the column-form extension evaluates
            # y_0 + m_0 (y_1 + m_1 (..)) — one product per word fewer than the sum over L_{i-1} y_i (ckks_ks.hip); the offset
            # of this second table travels in the descriptor (bits 16.. of the alpha word; 0 = no such table)
            horner = []
            for s_, (_, rows, primes) in enumerate(order):
                for i in range(len(primes)):
                    horner += [float(ctx.q[primes[i]] % ctx.q[r]) for r in dest]
            if self.ks_horner:
                for row in desc:
                    row[1] |= len(plain) << 16
            tabs[("extend", d)] = (self._t64(desc, d), self._t64(flat, d),
                                   torch.tensor(plain + horner, dtype=torch.float64, device=self.ntt.devices[d]))
            # (3) P_j^-1 R table, [K][rows]
            K, nrows = self.ntt.num_special_primes, len(dest)
            pir = torch.zeros((K, nrows), dtype=torch.int64, device=self.ntt.devices[d])
            for P_ind in range(K):
                v = self.PiRs[level][P_ind][d]
                pir[P_ind, :len(v)] = v
            tabs[("pir", d)] = pir
            # plain P_j^-1 mod q_row as doubles (fp64 class of the mod-down kernel)
            specials = ctx.q[-K:][::-1]
            # (5) the digit each local limb belongs to (255: special limbs) — cc_mult's key switch leaves those pairs out
            own = [255] * nrows
            for s_, (_, _, primes) in enumerate(order):
                for i, r in enumerate(dest):
                    if r in primes:
                        own[i] = s_
            tabs[("own", d)] = torch.tensor(own, dtype=torch.uint8, device=self.ntt.devices[d])
            tabs[("pip", d)] = torch.tensor(
                [[float(pow(specials[P_ind], -1, ctx.q[r])) if i < nrows - P_ind - 1 else 0.0 for i, r in enumerate(dest)]
                 for P_ind in range(K)], dtype=torch.float64, device=self.ntt.devices[d])
        # (4) exchange schedule (multi-device): runs of consecutive digits with the same owner travel as one
        # message — (owner, first digit, digits, first row in storage order, rows, first row in the owner's state)
        groups = []
        for s_, (d, rows, _) in enumerate(order):
            if groups and groups[-1][0] == d and groups[-1][5] + groups[-1][4] == rows[0]:
                g = groups[-1]
                groups[-1] = (g[0], g[1], g[2] + 1, g[3], g[4] + len(rows), g[5])
            else:
                groups.append((d, s_, 1, row_start[s_], len(rows), rows[0]))
        tabs["groups"] = groups
        self._tables[key] = tabs
        return tabs

    def _exchange_digits(self, states, level, tabs):
        """Every alive device receives every digit (eng.py:778-810: the reference stages all of them through pinned
        host memory first).  Returns {local device: (digits buffer [total_rows, N] in storage order,
        [(ready, first digit, digits), ...])}: the digits of a group may be read once `ready.wait()` has been
        called (ready None: already ordered on the current stream).  Buffers are allocated once per level; the
        list is in consumption order — one process per GPU: the digits this rank owns first, the foreign ones behind
        the single wait on the point-to-point batch."""
        n_alive = self.len_devices[level]
        loc = self._loc(level)
        nparts = len(tabs["order"])
        if n_alive == 1 and not (self._multi and self.comm.world_size == 1):
            return {loc[0]: (states[loc[0]], [(None, 0, nparts)])} if loc else {}
        N = self.ctx.N
        groups = tabs["groups"]
        if self._multi:
            me = self.local_ids[0]
            if not loc:   # no rows at this level: nothing to switch, nothing to send, nothing to receive ..
                if getattr(self.comm, "whole_group_exchange", False):   # .. unless the exchange is a whole-group collective
                    self.comm.exchange_rows(None, [(g[0], g[3], g[4]) for g in groups], list(range(n_alive)), width=N).wait()
                return {}
            buf = self._ws("ks_digits_all", (tabs["total_rows"], N), me)
            for owner, first, count, row0, nrows, src_row in groups:
                if owner == me:
                    buf[row0:row0 + nrows].copy_(states[me][src_row:src_row + nrows])
            # one batch of point-to-point messages between the alive ranks (comm.py); asynchronous
            handle = self.comm.exchange_rows(buf, [(g[0], g[3], g[4]) for g in groups], list(range(n_alive)))
            # consumption order: the digits this rank owns first (already here: their extension + NTT overlaps the
            # exchange), then — after ONE wait — the runs of foreign digits between them
            ready = [(None, first, count) for owner, first, count, _, _, _ in groups if owner == me]
            run = None
            for owner, first, count, _, _, _ in groups + [(me, nparts, 0, 0, 0, 0)]:
                if owner == me:
                    if run is not None:
                        ready.append((handle, run[0], run[1]))
                        handle, run = None, None
                else:
                    run = (first, count) if run is None else (run[0], run[1] + count)
            return {d: (buf, ready) for d in loc}
        out = {}
        for t in loc:
            buf = self._ws("ks_digits_all", (tabs["total_rows"], N), t)
            for owner, first, count, row0, nrows, src_row in groups:
                buf[row0:row0 + nrows].copy_(states[owner][src_row:src_row + nrows], non_blocking=True)
            out[t] = (buf, [(None, 0, nparts)])
        return out

    def create_switcher(self, a: list[torch.Tensor], ksk: data_struct, level, exit_ntt=False, addends=None,
                        galois=None, fold=None) -> tuple:
        """Key-switch the coefficient-domain polynomial `a` (one tensor per local device) under `ksk`.
        Returns (c0, c1) lists of canonical [rows, N] tensors.  `addends` = optional (list, list) added
        to (c0, c1) inside the last kernel (relinearize's d0/d1, switch_key's rotated c0).
        `galois` = (p^-1 mod 2N, canonical): switch a(X^p) and add addends(X^p) instead — the permutation is applied
        where the digits kernel and the mod-down kernel read their input (coefficient-domain `a` only).
        `fold` = {device: (x stack [4, ell, N], PR, own table)}: cc_mult's relinearisation — P * (x0 y0) and P * (x0 y1 + x1 y0) are
        added to the sums in the NTT domain (fused key switch only), see cc_mult."""
        tabs = self._ks_tables(level)
        loc = self._loc(level)
        N, logN, K = self.ctx.N, self.ctx.logN, self.ntt.num_special_primes
        packs = self._key_pack(ksk)
        loc0 = self._loc(0, special=True)

        # 1. mixed-radix digits of the local parts
        states = {}
        for i, d in enumerate(loc):
            src = a[i]
            rows = self._rows(d, level, False)
            if exit_ntt:
                src = src.clone()
                self.backend.intt(src, 1, rows, logN, self._tw(d, level, False, True), self._vec("Ninv", d, level, False),
                                  2, self._consts(d, level, False))
            st = self._ws("ks_state", (rows, N), d)
            nparts, desc, tab = tabs[("digits", d)]
            gal = None if galois is None else (galois[0], self._vec("_2q", d, level, False) if galois[1] else None)
            self.backend.ks_digits(src, st, nparts, desc, tab, self._consts(d, level, False), galois=gal)
            states[d] = st
        # 2. digit exchange: asynchronous, one batch of point-to-point messages between the alive ranks
        digits = self._exchange_digits(states, level, tabs)

        nparts = len(tabs["order"])
        c0, c1 = [], []
        for i, d in enumerate(loc):
            rows, ell = self._rows(d, level, True), self._rows(d, level, False)
            cs = self._consts(d, level, True)
            ext = self._ws("ks_ext", (nparts, rows, N), d)
            desc, E, Ed = tabs[("extend", d)]
            s = self._ws("ks_sum", (2, rows, N), d)
            key, tw, itw = packs[loc0.index(d)], self._tw(d, level, True), self._tw(d, level, True, True)
            ninv = self._vec("Ninv", d, level, True)
            dig, ready = digits[d]
            fused = logN >= self.backend.fused_ks_min_logN
            if fused and len(ready) > 1 and hasattr(self.backend, "ks_fwd"):
                # 3. extend + forward NTT of this rank's own digits while the others are on the wire, then of the
                # foreign runs; 4. once all are in: key inner product over all digits + inverse NTT
                for handle, first, count in ready:
                    if handle is not None:
                        handle.wait()
                    okw = {} if fold is None else {"own": fold[d][2]}
                    self.backend.ks_fwd(dig, first, count, rows, logN, desc, E, Ed, ext, tw, cs, **okw)
                fkw = {} if fold is None else {"fold": fold[d]}
                self.backend.ks_tail(nparts, rows, logN, key, tabs["first_part"], self.ntt.starts[level][d], ext, s, itw,
                                     ninv, cs, **fkw)
                ready = []
            for handle, _, _ in ready:
                if handle is not None:
                    handle.wait()
            if fused and ready:
                # 3+4. fused core: extend + NTT + key inner product + inverse NTT, the extended digits never
                # leave the chip in coefficient form
                fkw = {} if fold is None else {"fold": fold[d]}
                self.backend.ks_core(dig, nparts, rows, logN, desc, E, Ed, key, tabs["first_part"],
                                     self.ntt.starts[level][d], ext, s, tw, itw, ninv, cs, **fkw)
            elif not fused:
                assert fold is None
                # 3. extend every digit to this device's rows, forward NTT
                self.backend.ks_extend(dig, ext, nparts, rows, desc, E, cs)
                self.backend.ntt(ext, nparts, rows, logN, tw, None, cs, relaxed=True)
                # 4. inner product with the key (streams the key once), inverse NTT
                self.backend.ks_inner(ext, key, tabs["first_part"], self.ntt.starts[level][d], s[0], s[1], nparts, rows, cs)
                self.backend.intt(s, 2, rows, logN, itw, ninv, 2, cs, relaxed=True)
            # 5. divide by P (+ optional addend)
            out = torch.empty((2, ell, N), dtype=torch.int64, device=self.ntt.devices[d])
            rs = self._vec("Rs", d, level, True)
            adds = []
            for comp in range(2):
                add = addends[comp][i] if addends is not None and addends[comp] is not None else None
                if add is not None and not add.is_contiguous():
                    add = add.contiguous()
                adds.append(add)
            gal = None if galois is None else (galois[0], self._vec("_2q", d, level, False) if galois[1] else None)
            ws, one = self._moddown_ws("ks_moddown", 2, ell, K, d, tabs, cs)
            mkw = {"one_launch": True} if one else {}
            self.backend.ks_moddown_ws([s[0], s[1]], [out[0], out[1]], adds, ell, K, ws, tabs[("pir", d)], rs, cs,
                                       PiP=tabs[("pip", d)], galois=gal, **mkw)
            c0.append(out[0]); c1.append(out[1])
        return c0, c1

    # ---- the reference's key-switch steps as public methods (eng.py:218-227, 654-743, 906-937) ---------------------------
    # The reference's own callers are create_switcher and the multiparty helpers; create_switcher above runs fused
    # kernels instead.  Code written against the step methods still works: each one is the unfused C-ABI step of the
    # same name (lf_ks_digits / lf_ks_extend, lf_ntt, lf_mont_mult), so every intermediate word is the reference's.
    def reserve_ksk_buffers(self):
        """Staging buffers of the reference's digit gather (one pinned [K, N] host tensor per local part, eng.py:218-227).
        This engine moves digits device to device (comm.py) and never reads them; kept so that callers find the attribute."""
        K, N = self.ntt.num_special_primes, self.ctx.N
        self.ksk_buffers = []
        for d in range(self.ntt.num_devices):
            n_parts = len(self.ntt.p.p[0][d]) if d in self.local_ids else 0
            bufs = [torch.empty((K, N), dtype=torch.int64) for _ in range(n_parts)]
            if torch.cuda.is_available():
                bufs = [b.pin_memory() for b in bufs]
            self.ksk_buffers.append(bufs)

    def pre_extend(self, a, device_id, level, part_id, exit_ntt=False):
        """Mixed-radix (Garner) digits of part `part_id` of the polynomial a[device_id] ([rows, N]): the [alpha, N] state the
        reference returns, signed-lazy words included (eng.py:654-705)."""
        tabs = self._ks_tables(level)
        rows_of_part = self.ntt.p.parts[level][device_id][part_id]
        lo, alpha = rows_of_part[0], len(rows_of_part)
        i = self._loc(level).index(device_id) if len(a) != self.ntt.num_devices else device_id
        src = a[i]
        if exit_ntt:
            part = src[lo:lo + alpha]
            self.ntt.intt_exit_reduce([part], level, device_id, part_id)
        _, desc, tab = tabs[("digits", device_id)]
        one = desc[part_id:part_id + 1].clone()
        state = torch.empty((self._rows(device_id, level, False), self.ctx.N), dtype=torch.int64, device=src.device)
        self.backend.ks_digits(src.contiguous(), state, 1, one, tab, self._consts(device_id, level, False))
        return state[lo:lo + alpha]

    def extend(self, state, device_id, level, part_id, target_device_id=None):
        """Basis extension of a digit state (of part `part_id` of device `device_id`) to every limb, special ones included,
        of `target_device_id`: [rows, N] in Montgomery form (eng.py:707-743)."""
        target = device_id if target_device_id is None else target_device_id
        tabs = self._ks_tables(level)
        desc, E, _ = tabs[("extend", target)]
        s_id = self.stor_ids[level][device_id][part_id]
        one = desc[s_id:s_id + 1].clone()
        one[0, 0] = 0                                   # the state handed in starts at its own row 0
        rows = self._rows(target, level, True)
        st = state.to(self.ntt.devices[target]).contiguous()
        out = torch.empty((1, rows, self.ctx.N), dtype=torch.int64, device=st.device)
        self.backend.ks_extend(st, out, 1, rows, one, E, self._consts(target, level, True))
        return out[0]

    def switcher_later_part(self, state, ksk, src_device_id, dst_device_id, level, part_id):
        """extend -> NTT -> the two products with the key part (eng.py:906-937); returns (d0, d1) on dst_device_id."""
        extended = self.extend(state, src_device_id, level, part_id, dst_device_id)
        self.ntt.ntt([extended], level, dst_device_id, -2)
        part = ksk.data[self.parts_alloc[level][src_device_id][part_id]].data
        i = self._loc(0, special=True).index(dst_device_id)
        start = self.ntt.starts[level][dst_device_id]
        k0, k1 = part[0][i][start:], part[1][i][start:]
        d0 = self.ntt.mont_mult([extended], [k0], level, dst_device_id, -2)
        d1 = self.ntt.mont_mult([extended], [k1], level, dst_device_id, -2)
        return d0[0], d1[0]

    def switch_key(self, ct: data_struct, ksk: data_struct) -> data_struct:
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        level = ct.level
        c0, c1 = self.create_switcher(ct.data[1], ksk, level, exit_ntt=ct.ntt_state, addends=(ct.data[0], None))
        return data_struct(data=(c0, c1), include_special=ct.include_special, ntt_state=ct.ntt_state,
                           montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level, hash=self.hash)

    # =============================================================================================
    # rotation / conjugation (eng.py:1180-1263, 1718-1734)
    # =============================================================================================
    def _automorphism(self, ct, exponent, key, canonical):
        """X -> X^exponent on both components, then key-switch back.  `canonical`: fold the reference's
        make_unsigned + reduce_2q (rotate_single does that, eng.py:1198-1200; conjugate does not and
        key-switches the signed words, eng.py:1718-1734).
        Coefficient-domain ciphertexts (the normal case) never see a permutation pass: the key switch reads
        c1(X^p) and adds c0(X^p) in gather form inside its first and last kernels."""
        level = ct.level
        if not ct.ntt_state and not ct.include_special:
            pinv = pow(exponent, -1, 2 * self.ctx.N)
            d = self._native_level(level)
            if d is not None and ct.data[0][0].is_contiguous() and ct.data[1][0].is_contiguous():
                # the whole rotation as ONE native call (lf_switch_key)
                plan, _, first_part, row_off = self._op_plan(level, d)
                kpack = self._key_pack(key)[self._loc(0, special=True).index(d)]
                out = torch.empty((2, plan.ell, self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
                self.backend.switch_key_native(plan, ct.data[0][0], ct.data[1][0], pinv, canonical, kpack, first_part, row_off, out)
                return data_struct(data=([out[0]], [out[1]]), include_special=False, ntt_state=False,
                                   montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level, hash=self.hash)
            d = self._sharded_native(level)
            if d is not None and ct.data[0][0].is_contiguous() and ct.data[1][0].is_contiguous():
                # a limb-sharded level: digits, [exchange], extension + NTT per run, tail — the host side in native calls
                plan, _, first_part, row_off = self._op_plan(level, d)
                kpack = self._key_pack(key)[self._loc(0, special=True).index(d)]
                out = torch.empty((2, plan.ell, self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
                self.backend.switch_key_pre(plan, ct.data[1][0], pinv, canonical)     # one launch, reads c1
                seg = self._sharded_segments("switch", level, d, plan, kpack, first_part, row_off)
                if seg is not None:
                    self._sharded_exchange_replay(seg)
                    self.backend.switch_key_post(plan, ct.data[0][0], pinv, canonical, kpack, first_part, row_off, out, which=2)
                else:
                    self._sharded_forward(plan, d, level, False)
                    self.backend.switch_key_post(plan, ct.data[0][0], pinv, canonical, kpack, first_part, row_off, out)
                    self._sharded_segments("switch", level, d, plan, kpack, first_part, row_off, capture=True)   # the next op replays
                return data_struct(data=([out[0]], [out[1]]), include_special=False, ntt_state=False,
                                   montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level, hash=self.hash)
            c0, c1 = self.create_switcher(ct.data[1], key, level, addends=(ct.data[0], None), galois=(pinv, canonical))
            return data_struct(data=(c0, c1), include_special=False, ntt_state=False,
                               montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level, hash=self.hash)
        loc = self._loc(level, special=ct.include_special)
        rot0, rot1 = [], []
        for i, d in enumerate(loc):
            rows = ct.data[0][i].size(0)
            r = torch.empty((2, rows, self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
            _2q = self._vec("_2q", d, level, ct.include_special) if canonical else None
            srcs = [t if t.is_contiguous() else t.contiguous() for t in (ct.data[0][i], ct.data[1][i])]
            self.backend.galois_batch(srcs, [r[0], r[1]], rows, self.ctx.logN, exponent, _2q)
            rot0.append(r[0]); rot1.append(r[1])
        permuted = data_struct(data=(rot0, rot1), include_special=ct.include_special, ntt_state=ct.ntt_state,
                               montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level,
                               hash=self.hash, version=self.version)
        return self.switch_key(permuted, key)

    def rotate_single(self, ct: data_struct, rotk: data_struct) -> data_struct:
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        if types.origins["rotk"] not in rotk.origin:
            raise errors.NotMatchType(origin=rotk.origin, to=types.origins["rotk"])
        delta = int(rotk.origin.split(":")[-1])
        return self._automorphism(ct, encdec.galois_exponent(self.ctx.N, delta), rotk, canonical=True)

    def rotate_single_batch(self, cts: list, rotk: data_struct) -> list:
        """rotate_single of several ciphertexts by the same step, i.e. under the same key (BASELINE config "rotate
        batched 64 ciphertexts"; the reference has no batched entry and loops).  Returns
        [rotate_single(ct, rotk) for ct in cts], bit for bit.  On one GPU, groups of up to 4 coefficient-domain
        ciphertexts of one level go through the key switch together: one launch set per group (4x the blocks per
        launch) and the key — two thirds of the inner product's HBM bytes — is read once per group."""
        if types.origins["rotk"] not in rotk.origin:
            raise errors.NotMatchType(origin=rotk.origin, to=types.origins["rotk"])
        for ct in cts:
            if ct.origin != types.origins["ct"]:
                raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        out = [None] * len(cts)
        sizes = getattr(self.backend, "ks_batch_sizes", ())
        groups = {}
        for i, ct in enumerate(cts):
            ok = (sizes and not ct.ntt_state and not ct.include_special and self.len_devices[ct.level] == 1
                  and len(self._loc(ct.level)) == 1 and self.ctx.logN >= self.backend.fused_ks_min_logN)
            if ok:
                groups.setdefault(ct.level, []).append(i)
            else:
                out[i] = self.rotate_single(ct, rotk)
        delta = int(rotk.origin.split(":")[-1])
        exponent = encdec.galois_exponent(self.ctx.N, delta)
        for level, idx in groups.items():
            jobs, slots, pos = [], [], 0
            while pos < len(idx):
                n = next((k for k in sizes if k <= len(idx) - pos), 1)
                sel = idx[pos:pos + n]
                if n == 1:
                    jobs.append(lambda sel=sel: [self.rotate_single(cts[sel[0]], rotk)])
                else:
                    jobs.append(lambda sel=sel: self._automorphism_batch([cts[i] for i in sel], exponent, rotk, level))
                slots.append(sel)
                pos += n
            for sel, res in zip(slots, self._run_groups(jobs, self._loc(level)[0], lambda: self._prepare_ks(rotk, level))):
                for i, r in zip(sel, res):
                    out[i] = r
        return out

    def _prepare_ks(self, key, level):
        """Build, on the CURRENT stream, every lazily built piece of shared state a key switch at `level` under
        `key` reads: the packed key, the per-level descriptor tables, the fp64 twins of the twiddle tables."""
        self._key_pack(key)
        self._ks_tables(level)
        warm = getattr(self.backend, "warm_twiddles", None)
        if warm is not None:
            for d in self._loc(level):
                for special in (False, True):
                    c = self._consts(d, level, special)
                    warm(self._tw(d, level, special), c)
                    warm(self._tw(d, level, special, True), c)

    def _run_groups(self, jobs, dev_id, prepare=None):
        """Run independent group jobs (callables returning lists of data_structs) alternately on two streams of
        the device (LF_ENGINE_LANES, Python side only; measured: 2 beats 1 and 3): a group's memory-bound phases (the key stream of the inner product) overlap the other group's
        instruction-bound ones, and launch tails fill.  Each lane has its own workspaces.  Returns the results in
        order; the caller's stream waits for the side lane before returning.
        Lazily built shared state (key pack, per-level tables, fp64 twiddle twins) is produced by launches on the
        stream of whichever call touches it first: `prepare()` builds all of it on the caller's stream BEFORE the
        side lane is forked (idempotent: dictionary hits afterwards), so the side lane never reads it half-built."""
        device = self.ntt.devices[dev_id]
        lanes = max(1, int(os.environ.get("LF_ENGINE_LANES", "2")))
        if len(jobs) < 2 or not str(device).startswith("cuda") or lanes == 1:
            return [job() for job in jobs]
        main = torch.cuda.current_stream(device)
        sides = self._lane_streams.setdefault(dev_id, [])
        while len(sides) < lanes - 1:
            sides.append(torch.cuda.Stream(device=device))
        results = []
        if prepare is not None:
            prepare()
        fork = torch.cuda.Event()
        fork.record(main)
        for side in sides[:lanes - 1]:
            side.wait_event(fork)
        try:
            for n, job in enumerate(jobs):
                self._lane = n % lanes
                if self._lane:
                    with torch.cuda.stream(sides[self._lane - 1]):
                        res = job()
                    for r in res:
                        for comp in r.data:
                            for t in comp:
                                t.record_stream(main)
                else:
                    res = job()
                results.append(res)
        finally:
            self._lane = 0
        for side in sides[:lanes - 1]:
            join = torch.cuda.Event()
            join.record(side)
            main.wait_event(join)
        return results

    def _automorphism_batch(self, cts, exponent, key, level):
        """X -> X^exponent + key switch of len(cts) in (2, 4) ciphertexts of one level on the single local device
        (the rotate_single form: canonical words)."""
        d = self._loc(level)[0]
        pinv = pow(exponent, -1, 2 * self.ctx.N)
        if self._native_level(level) == d and hasattr(self.backend, "switch_key_batch_native") and \
                all(ct.data[0][0].is_contiguous() and ct.data[1][0].is_contiguous() for ct in cts):
            # the whole group as ONE native call (lf_switch_key_batch)
            plan, _, first_part, row_off = self._op_plan(level, d, len(cts))
            kpack = self._key_pack(key)[self._loc(0, special=True).index(d)]
            out = torch.empty((len(cts), 2, plan.ell, self.ctx.N), dtype=torch.int64, device=self.ntt.devices[d])
            self.backend.switch_key_batch_native(plan, [ct.data[0][0] for ct in cts], [ct.data[1][0] for ct in cts], pinv, True, kpack,
                                                 first_part, row_off, out)
        else:
            gal = (pinv, self._vec("_2q", d, level, False))
            out = self._ks_batch([ct.data[1][0] for ct in cts], [(ct.data[0][0], None) for ct in cts], key, level, gal)
        return [data_struct(data=([out[b][0]], [out[b][1]]), include_special=False, ntt_state=False,
                            montgomery_state=ct.montgomery_state, origin=types.origins["ct"], level=level, hash=self.hash)
                for b, ct in enumerate(cts)]

    def _ks_batch(self, srcs, addends, key, level, gal=None, fold=None):
        """Key switch of len(srcs) in (2, 4) coefficient-domain polynomials ([ell, N] tensors on the single local
        device of `level`) under one key; addends[b] = (add to c0, add to c1) or Nones.  Returns [nct, 2, ell, N].
        fold = (x [nct, 4, ell, N], PR, own): see create_switcher."""
        tabs = self._ks_tables(level)
        d = self._loc(level)[0]
        N, logN, K = self.ctx.N, self.ctx.logN, self.ntt.num_special_primes
        nct = len(srcs)
        rows, ell = self._rows(d, level, True), self._rows(d, level, False)
        c_ord, cs = self._consts(d, level, False), self._consts(d, level, True)
        # 1. mixed-radix digits of all polynomials, one launch, into a common stack
        states = self._ws("ks_state_batch", (nct, ell, N), d)
        nparts_d, desc_d, tab_d = tabs[("digits", d)]
        srcs = [src if src.is_contiguous() else src.contiguous() for src in srcs]
        self.backend.ks_digits_batch(srcs, [states[b] for b in range(nct)], nparts_d, desc_d, tab_d, c_ord, galois=gal)
        # 2. fused core over the whole group
        nparts = len(tabs["order"])
        ext = self._ws("ks_ext_batch", (nct, nparts, rows, N), d)
        s = self._ws("ks_sum_batch", (nct, 2, rows, N), d)
        desc, E, Ed = tabs[("extend", d)]
        packs = self._key_pack(key)
        kpack = packs[self._loc(0, special=True).index(d)]
        fkw = {} if fold is None else {"fold": fold}
        self.backend.ks_core_batch(states, nparts, rows, logN, desc, E, Ed, kpack, tabs["first_part"],
                                   self.ntt.starts[level][d], ext, s, self._tw(d, level, True),
                                   self._tw(d, level, True, True), self._vec("Ninv", d, level, True), cs, **fkw)
        # 3. divide by P (+ addends, in gather form under a Galois map): one launch pair for the 2 * nct polynomials
        out = torch.empty((nct, 2, ell, N), dtype=torch.int64, device=self.ntt.devices[d])
        ss = [s[b][comp] for b in range(nct) for comp in range(2)]
        outs = [out[b][comp] for b in range(nct) for comp in range(2)]
        adds = []
        for pair in addends:
            for a in pair:
                adds.append(a if a is None or a.is_contiguous() else a.contiguous())
        ws, one = self._moddown_ws("ks_moddown_batch", 2 * nct, ell, K, d, tabs, cs)
        mkw = {"one_launch": True} if one else {}
        self.backend.ks_moddown_ws(ss, outs, adds, ell, K, ws, tabs[("pir", d)], self._vec("Rs", d, level, True), cs,
                                   PiP=tabs[("pip", d)], galois=gal, **mkw)
        return out

    def cc_mult_batch(self, pairs: list, evk: data_struct) -> list:
        """cc_mult (+ relinearize) of several ciphertext pairs under one evaluation key: returns
        [cc_mult(a, b, evk) for a, b in pairs], bit for bit.  On one GPU, groups of up to 4 pairs of one level share
        the inverse transforms of their triplets (one launch) and the key switch (lf_ks_core_batch: one launch set,
        the key read once per group)."""
        out = [None] * len(pairs)
        sizes = getattr(self.backend, "ks_batch_sizes", ())
        groups = {}
        for i, (a, b) in enumerate(pairs):
            if a.origin != types.origins["ct"] or b.origin != types.origins["ct"]:
                raise errors.NotMatchType(origin=f"{a.origin} and {b.origin}", to=types.origins["ct"])
            lvl = a.level + 1
            ok = (sizes and a.level == b.level and lvl < self.num_levels and self.len_devices[lvl] == 1
                  and len(self._loc(lvl)) == 1 and self.len_devices[a.level] == 1
                  and self.ctx.logN >= self.backend.fused_ks_min_logN
                  and not (a.ntt_state or b.ntt_state or a.include_special or b.include_special))
            if ok:
                groups.setdefault(lvl, []).append(i)
            else:
                out[i] = self.cc_mult(a, b, evk)
        for level, idx in groups.items():
            jobs, slots, pos = [], [], 0
            while pos < len(idx):
                n = next((k for k in sizes if k <= len(idx) - pos), 1)
                sel = idx[pos:pos + n]
                if n == 1:
                    jobs.append(lambda sel=sel: [self.cc_mult(pairs[sel[0]][0], pairs[sel[0]][1], evk)])
                else:
                    jobs.append(lambda sel=sel: self._cc_mult_group([pairs[i] for i in sel], evk, level))
                slots.append(sel)
                pos += n
            for sel, res in zip(slots, self._run_groups(jobs, self._loc(level)[0], lambda: self._prepare_ks(evk, level))):
                for i, r in zip(sel, res):
                    out[i] = r
        return out

    def _cc_mult_group(self, pairs, evk, level):
        d = self._loc(level)[0]
        N, logN = self.ctx.N, self.ctx.logN
        nct = len(pairs)
        rows = self._rows(d, level, False)
        if self._native_level(level) == d and self._native_level(level - 1) == d and hasattr(self.backend, "cc_mult_evk_batch_native") \
                and getattr(self.backend, "relin_fold", False) \
                and all(t.is_contiguous() and t.dtype == torch.int64 for a, b in pairs for ct in (a, b) for t in (ct.data[0][0], ct.data[1][0])):
            # the whole group as ONE native call (lf_cc_mult_evk_batch)
            plan, _, first_part, row_off = self._op_plan(level, d, nct)
            ins, row0s = (ctypes.c_void_p * (4 * nct))(), (ctypes.c_void_p * (4 * nct))()
            k = 0
            for a, b in pairs:
                for ct in (a, b):
                    for comp in range(2):
                        ptr = ct.data[comp][0].data_ptr()
                        row0s[k], ins[k] = ptr, ptr + N * 8      # the dropped limb is the first row; the survivors follow it
                        k += 1
            kpack = self._key_pack(evk)[self._loc(0, special=True).index(d)]
            out = torch.empty((nct, 2, plan.ell, N), dtype=torch.int64, device=self.ntt.devices[d])
            self.backend.cc_mult_evk_batch_native(plan, ins, row0s, kpack, first_part, row_off, out)
            return [self._new(([out[t][0]], [out[t][1]]), types.origins["ct"], level=level) for t in range(nct)]
        c = self._consts(d, level, False)
        fold = getattr(self.backend, "relin_fold", False)
        # rescale + forward transform of the operands, two pairs (8 polynomials) per launch
        x = self._ws("multx", (nct, 4, rows, N), d) if fold else self._ws("mult8", (8, rows, N), d)
        trip = None if fold else torch.empty((nct, 3, rows, N), dtype=torch.int64, device=self.ntt.devices[d])
        for t0 in range(0, nct, 2):
            chunk = pairs[t0:t0 + 2]
            per_dev, round_at = self._rescale_operands([ct for pair in chunk for ct in pair])
            srcs, r0s = per_dev[d]
            xs = x[t0:t0 + len(chunk)].view(4 * len(chunk), rows, N) if fold else x[:4 * len(chunk)]
            self.backend.rescale_ntt(srcs, r0s, xs, rows, logN, self.rescale_scales[level - 1][d], round_at,
                                     self._tw(d, level, False), self._vec("Rs", d, level, False), c, relaxed=True, plain=True)
            if fold:
                continue
            for k in range(len(chunk)):
                t = t0 + k
                self.backend.tensor(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3], trip[t][0], trip[t][1], trip[t][2],
                                    rows, c, plain=True)
        if fold:
            # the nct products x1 * y1 through one inverse transform (product on load); d0 / d1 folded into the key switch
            d2 = self._ws("multx_d2", (nct, rows, N), d)
            self.backend.intt_mul(d2, x[0][1], x[0][3], nct, rows, logN, self._tw(d, level, False, True),
                                  self._vec("Ninv", d, level, False), c, a_stride=4 * rows * N, b_stride=4 * rows * N)
            out = self._ks_batch([d2[t] for t in range(nct)], [(None, None)] * nct, evk, level,
                                 fold=(x, self._PR(d, level), self._ks_tables(level)[("own", d)]))
            return [self._new(([out[t][0]], [out[t][1]]), types.origins["ct"], level=level) for t in range(nct)]
        # the 3 * nct inverse transforms of the triplets: one launch
        self.backend.intt(trip.view(3 * nct, rows, N), 3 * nct, rows, logN, self._tw(d, level, False, True),
                          self._vec("Ninv", d, level, False), 2, c, relaxed=True, plain=True)
        out = self._ks_batch([trip[t][2] for t in range(nct)], [(trip[t][0], trip[t][1]) for t in range(nct)], evk, level)
        return [self._new(([out[t][0]], [out[t][1]]), types.origins["ct"], level=level) for t in range(nct)]

    def rotate_galois(self, ct: data_struct, gk: data_struct, delta: int, return_circuit=False) -> data_struct:
        if ct.origin != types.origins["ct"]:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        if gk.origin != types.origins["galk"]:
            raise errors.NotMatchType(origin=gk.origin, to=types.origins["galk"])
        remaining = delta % (self.ctx.N // 2)
        circuit = []
        while remaining:
            ind = int(math.log2(remaining))
            circuit.append(ind)
            remaining -= self.galois_deltas[ind]
        out = ct
        for ind in circuit:
            out = self.rotate_single(out, gk.data[ind])
        return (out, circuit) if return_circuit else out

    def conjugate(self, ct: data_struct, conjk: data_struct):
        return self._automorphism(ct, encdec.conjugation_exponent(self.ctx.N), conjk, canonical=False)

    # =============================================================================================
    # add / sub (eng.py:1268-1405)
    # =============================================================================================
    def _cc_linear(self, a, b, op, want):
        if a.origin != types.origins[want] or b.origin != types.origins[want]:
            raise errors.NotMatchType(origin=f"{a.origin} and {b.origin}", to=types.origins[want])
        lazy = want == "ctt"
        for x in (a, b):
            if (x.ntt_state, x.montgomery_state) != (lazy, lazy):
                raise errors.NotMatchDataStructState(origin=x.origin)
        level = a.level
        fn = self.ntt.mont_add if op == "add" else self.ntt.mont_sub
        data = []
        for xa, xb in zip(a.data, b.data):
            c = fn(xa, xb, level)
            self.ntt.reduce_2q(c, level)
            data.append(c)
        return self._new(data, types.origins[want], level=level, ntt_state=lazy, montgomery_state=lazy)

    def cc_add_double(self, a, b):
        return self._cc_linear(a, b, "add", "ct")

    def cc_add_triplet(self, a, b):
        return self._cc_linear(a, b, "add", "ctt")

    def cc_sub_double(self, a, b):
        return self._cc_linear(a, b, "sub", "ct")

    def cc_sub_triplet(self, a, b):
        return self._cc_linear(a, b, "sub", "ctt")

    def cc_add(self, a: data_struct, b: data_struct) -> data_struct:
        if a.origin == types.origins["ct"] and b.origin == types.origins["ct"]:
            return self.cc_add_double(a, b)
        if a.origin == types.origins["ctt"] and b.origin == types.origins["ctt"]:
            return self.cc_add_triplet(a, b)
        raise errors.DifferentTypeError(a=a.origin, b=b.origin)

    def cc_sub(self, a: data_struct, b: data_struct) -> data_struct:
        if a.origin != b.origin:
            raise Exception("[Error] triplet error")
        if a.origin == types.origins["ct"]:
            return self.cc_sub_double(a, b)
        if a.origin == types.origins["ctt"]:
            return self.cc_sub_triplet(a, b)
        raise errors.DifferentTypeError(a=a.origin, b=b.origin)

    def cc_subtract(self, a, b):
        return self.cc_sub(a, b)

    # =============================================================================================
    # level management and type-dispatched operators (eng.py:1410-1467, 2225-2283)
    # =============================================================================================
    def level_up(self, ct: data_struct, dst_level: int):
        if types.origins["ct"] != ct.origin:
            raise errors.NotMatchType(origin=ct.origin, to=types.origins["ct"])
        new_ct = self.rescale(ct)
        src_level = ct.level + 1
        delta = round(self.scale * (self.deviations[dst_level] / np.sqrt(self.deviations[src_level])))
        loc_src, loc_dst = self._loc(src_level), self._loc(dst_level)
        data = []
        for comp in range(2):
            rows = []
            for d in loc_dst:
                t = new_ct.data[comp][loc_src.index(d)]
                rows.append(t[t.size(0) - self._rows(d, dst_level, False):].clone())
            data.append(rows)
        mult = [self._t64([delta * self.ctx.R % self.ctx.q[i] for i in self.ntt.p.destination_arrays[dst_level][d]], d)
                for d in loc_dst]
        for comp in range(2):
            self.ntt.mont_enter_scalar(data[comp], mult, dst_level)
            self.ntt.reduce_2q(data[comp], dst_level)
        return self._new(tuple(data), types.origins["ct"], level=dst_level)

    def auto_level(self, ct0, ct1):
        l0, l1 = ct0.level, ct1.level
        if l0 < l1:
            return self.level_up(ct0, l1), ct1
        if l0 > l1:
            return ct0, self.level_up(ct1, l0)
        return ct0, ct1

    def auto_cc_mult(self, ct0, ct1, evk, relin=True):
        a, b = self.auto_level(ct0, ct1)
        return self.cc_mult(a, b, evk, relin=relin)

    def auto_cc_add(self, ct0, ct1):
        a, b = self.auto_level(ct0, ct1)
        return self.cc_add(a, b)

    def auto_cc_sub(self, ct0, ct1):
        a, b = self.auto_level(ct0, ct1)
        return self.cc_sub(a, b)

    def _dispatch(self, table, a, b):
        try:
            return table[data_struct if is_struct(a) else type(a), data_struct if is_struct(b) else type(b)]
        except Exception as e:
            raise Exception(f"Unsupported data types are input.\n{e}")

    def mult(self, a, b, evk=None, relin=True):
        return self._dispatch(self.mult_dispatch_dict, a, b)(a, b, evk, relin)

    def add(self, a, b):
        return self._dispatch(self.add_dispatch_dict, a, b)(a, b)

    def sub(self, a, b):
        return self._dispatch(self.sub_dispatch_dict, a, b)(a, b)

    def square(self, ct: data_struct, evk: data_struct, relin=True) -> data_struct:
        return self.cc_mult(ct, ct, evk, relin=relin)
