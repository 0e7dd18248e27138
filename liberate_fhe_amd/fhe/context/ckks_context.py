"""RNS-CKKS parameter context: prime chain, Montgomery constants and NTT twiddle tables.

Host-side mirror of the reference's `ckks_context` (src/liberate/fhe/context/ckks_context.py:150-341):
same constructor keywords, same attribute names and values (`q`, `R`, `R_square`, `q_double`,
`q_lower_bits`, `k`, `N_inv`, ...).  What differs is the shape of the twiddle data: the reference
materialises `[limbs, logN, N/2]` per-stage tables plus `[logN, N/2]` gather-index tables (164 MB for
the gold preset); the HIP kernels index a compact `[limbs, N]` bit-reversed power table instead
(`psi_br[x] = psi^brev(x)`), so those are what this context builds.  The per-stage tables are still
available (`forward_psi`, `forward_even_indices`, ...) for callers written against the reference, and
are derived lazily from the compact tables.
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch

from ..presets import errors
from . import primes as _primes
from .security import maximum_qbits


def bit_reverse_indices(logN: int) -> np.ndarray:
    """brev_logN(i) for i < 2^logN, built by doubling (no string round-trips)."""
    rev = np.zeros(1, dtype=np.int64)
    for _ in range(logN):
        rev = np.concatenate([2 * rev, 2 * rev + 1])
    return rev


def primitive_root_2N(q: int, N: int) -> int:
    """First x^((q-1)/2N), x = 2, 3, ..., whose N-th power is not 1 (ckks_context.py:20-28)."""
    e = (q - 1) // (2 * N)
    for x in range(2, N):
        g = pow(x, e, q)
        if pow(g, N, q) != 1:
            return g
    raise ValueError(f"no primitive 2N-th root found for q={q}")


def _power_table(g: int, N: int, q: int) -> np.ndarray:
    out = np.empty(N, dtype=np.int64)
    v = 1
    for i in range(N):
        out[i] = v
        v = v * g % q
    return out


def stage_butterfly_indices(logN: int, inverse: bool):
    """Gather indices and twiddle indices of every radix-2 stage, in the reference's launch order.

    forward (ckks_context.py:89-112): stage s, m = 2^s, t = N/2m; butterfly b -> block i = b // t,
    U = 2*i*t + b % t, V = U + t, twiddle index m + i.
    inverse (ckks_context.py:115-142): stage s, t = 2^s, h = N/2t; same (i, U, V), twiddle h + i.
    """
    N = 1 << logN
    b = np.arange(N // 2, dtype=np.int64)
    even = np.empty((logN, N // 2), dtype=np.int32)
    odd = np.empty((logN, N // 2), dtype=np.int32)
    tw = np.empty((logN, N // 2), dtype=np.int32)
    for s in range(logN):
        t = (1 << s) if inverse else (N >> (s + 1))
        base = (N >> (s + 1)) if inverse else (1 << s)
        i = b // t
        U = 2 * i * t + b % t
        even[s], odd[s], tw[s] = U, U + t, base + i
    return even, odd, tw


class ckks_context:
    @errors.log_error
    def __init__(
        self,
        buffer_bit_length=62,
        scale_bits=40,
        logN=15,
        num_scales=None,
        num_special_primes=2,
        sigma=3.2,
        uniform_ternary_secret=True,
        cache_folder=None,
        security_bits=128,
        quantum="post_quantum",
        distribution="uniform",
        read_cache=True,
        save_cache=True,
        verbose=False,
        is_secured=True,
    ):
        # cache_folder / read_cache / save_cache are accepted for call compatibility; every table
        # here is regenerated in about a second, so nothing is written to disk.
        if buffer_bit_length not in (30, 62):
            raise errors.NotFindBufferBitLength(buffer_bit_length)
        # 30: the reference's int32 word mode (ckks_context.py:213-216; R = 2^30, 28-bit message primes).  The contexts and
        # the 15 ntt_cuda functions serve it (csrc/ckks_w30.hip); the engine's fused ops are 62-bit only.

        self.generation_string = (
            f"{buffer_bit_length}_{scale_bits}_{logN}_{num_scales}_"
            f"{num_special_primes}_{security_bits}_{quantum}_{distribution}"
        )
        self.is_secured = is_secured
        self.buffer_bit_length = buffer_bit_length
        self.scale_bits = scale_bits
        self.logN = logN
        self.num_special_primes = num_special_primes
        self.cache_folder = cache_folder
        self.security_bits = security_bits
        self.quantum = quantum
        self.distribution = distribution
        self.sigma = sigma
        self.uniform_ternary_secret = uniform_ternary_secret
        self.secret_key_sampling_method = "uniform ternary" if uniform_ternary_secret else "sparse ternary"
        self.torch_dtype = {30: torch.int32, 62: torch.int64}[buffer_bit_length]
        self.numpy_dtype = {30: np.int32, 62: np.int64}[buffer_bit_length]
        self.N = 1 << logN
        self.message_bits = buffer_bit_length - 2

        if not 12 <= logN <= 17:
            raise errors.NotFoundMessageSpecialPrimes(message_bit=self.message_bits, N=self.N)
        message_special = list(_primes.message_special_primes(self.message_bits, self.N))
        try:
            pool = list(_primes.scale_prime_pool(scale_bits, self.N))
        except Exception:
            raise errors.NotFoundScalePrimes(scale_bits=scale_bits, N=self.N)

        # chain = [scale primes (dropped left to right by rescale), base prime, special primes]
        self.max_qbits = int(maximum_qbits(self.N, security_bits, quantum, distribution))
        base_special = message_special[: 1 + num_special_primes]
        try:
            if num_scales is None:
                budget = self.max_qbits - sum(math.log2(p) for p in base_special)
                num_scales = 0
                budget -= math.log2(pool[num_scales])
                while budget > 0:
                    num_scales += 1
                    budget -= math.log2(pool[num_scales])
            if num_scales > len(pool):
                raise IndexError
            self.num_scales = num_scales
            self.q = pool[:num_scales] + base_special
        except IndexError:
            raise errors.NotEnoughPrimes(scale_bits=scale_bits, N=self.N)

        self.total_qbits = math.ceil(sum(math.log2(qi) for qi in self.q))
        if self.total_qbits > self.max_qbits:
            if is_secured:
                raise errors.ViolatedAllowedQbits(
                    scale_bits=scale_bits, N=self.N, num_scales=self.num_scales,
                    max_qbits=self.max_qbits, total_qbits=self.total_qbits)
            warnings.warn(
                f"Maximum allowed qbits are violated: max_qbits={self.max_qbits:4d} and the "
                f"requested total is {self.total_qbits:4d}."
            )

        self.generate_montgomery_parameters()
        self.generate_paints()
        if verbose:
            self.init_print()

    # -- Montgomery constants (ckks_context.py:294-315) -----------------------------------------
    def generate_montgomery_parameters(self):
        self.R = 1 << self.buffer_bit_length
        self.half_buffer_bit_length = self.buffer_bit_length // 2
        self.lower_bits_mask = (1 << self.half_buffer_bit_length) - 1
        self.full_bits_mask = (1 << self.buffer_bit_length) - 1
        h, lo = self.half_buffer_bit_length, self.lower_bits_mask
        self.R_square = [self.R * self.R % qi for qi in self.q]
        self.q_lower_bits = [qi & lo for qi in self.q]
        self.q_higher_bits = [qi >> h for qi in self.q]
        self.q_double = [2 * qi for qi in self.q]
        self.R_inv = [pow(self.R, -1, qi) for qi in self.q]
        # k*q == -1 (mod R)
        self.k = [(self.R * ri - 1) // qi for ri, qi in zip(self.R_inv, self.q)]
        self.k_lower_bits = [ki & lo for ki in self.k]
        self.k_higher_bits = [ki >> h for ki in self.k]

    # -- twiddles --------------------------------------------------------------------------------
    def generate_paints(self):
        self.N_inv = [pow(self.N, -1, qi) for qi in self.q]
        brev = bit_reverse_indices(self.logN)
        self.psi_root = [primitive_root_2N(qi, self.N) for qi in self.q]
        # psi_br[l, x] = psi_l^brev(x) mod q_l ; ipsi_br likewise with psi^-1 (plain residues).
        self.psi_br = np.stack([_power_table(g, self.N, qi)[brev] for g, qi in zip(self.psi_root, self.q)])
        self.ipsi_br = np.stack(
            [_power_table(pow(g, -1, qi), self.N, qi)[brev] for g, qi in zip(self.psi_root, self.q)])
        self._stage_tables = {}

    def _stages(self, inverse):
        if inverse not in self._stage_tables:
            self._stage_tables[inverse] = stage_butterfly_indices(self.logN, inverse)
        return self._stage_tables[inverse]

    # Reference-shaped views (ckks_context.py:317-341), built on first use.
    @property
    def forward_even_indices(self):
        return self._stages(False)[0]

    @property
    def forward_odd_indices(self):
        return self._stages(False)[1]

    @property
    def backward_even_indices(self):
        return self._stages(True)[0]

    @property
    def backward_odd_indices(self):
        return self._stages(True)[1]

    @property
    def forward_psi(self):
        return self.psi_br[:, self._stages(False)[2]]

    @property
    def backward_psi_inv(self):
        return self.ipsi_br[:, self._stages(True)[2]]

    def init_print(self):
        print(f"""
I have received inputs:
        buffer_bit_length\t\t= {self.buffer_bit_length:,d}
        scale_bits\t\t\t= {self.scale_bits:,d}
        logN\t\t\t\t= {self.logN:,d}
        N\t\t\t\t= {self.N:,d}
        Number of special primes\t= {self.num_special_primes:,d}
        Number of scales\t\t= {self.num_scales:,d}
        Cache folder\t\t\t= '{self.cache_folder}'
        Security bits\t\t\t= {self.security_bits:,d}
        Quantum security model\t\t= {self.quantum:s}
        Security sampling distribution\t= {self.distribution:s}
        Number of message bits\t\t= {self.message_bits:,d}
        In total I will be using '{self.total_qbits:,d}' bits out of available maximum '{self.max_qbits:,d}' bits.
        And is it secured?\t\t= {self.is_secured}
My RNS primes are {self.q}.""")
