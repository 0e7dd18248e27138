"""Random samplers behind the engine's keygen / encrypt (interface of the reference's `Csprng`).

SCOPE NOTE: the reference samples with fused ChaCha20 CUDA kernels
(src/liberate/csprng/*.cu).  Sampling is not ciphertext arithmetic and is outside this round's hot
path; this class keeps the reference's method names, argument meaning, shapes, dtypes and value
ranges (csprng.py:18-323) on top of torch's device generator (Philox).  It is NOT yet the ChaCha20
CSPRNG and must not be used where cryptographic randomness is required.

Invariant kept from the reference (csprng.py:146-158): the `repeats` trailing channels — used for the
special-prime rows that every GPU replicates — are drawn from a stream that is identical on every
GPU / rank, the leading channels from per-GPU streams.
"""
from __future__ import annotations

import os

import numpy as np
import torch


class Csprng:
    def __init__(self, num_coefs=2 ** 15, num_channels=(8,), num_repeating_channels=2, sigma=3.2,
                 devices=None, seed=None, nonce=None, shared_seed=None, local_ids=None):
        self.num_coefs = num_coefs
        self.num_channels = list(num_channels)
        self.num_repeating_channels = num_repeating_channels
        self.sigma = sigma
        if devices is None:
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        self.devices = list(devices)
        self.num_devices = len(self.devices)
        if len(self.num_channels) == 1:
            self.shares = self.num_channels * self.num_devices
        elif len(self.num_channels) == self.num_devices:
            self.shares = self.num_channels
        else:
            raise Exception("There was a contradicting mismatch between num_channels, and devices.")
        self.local_ids = list(range(self.num_devices)) if local_ids is None else list(local_ids)
        if shared_seed is None:
            shared_seed = int.from_bytes(os.urandom(7), "little")
        self._shared_seed = shared_seed
        self._own, self._shared = {}, {}
        for dev_id in self.local_ids:
            dev = self.devices[dev_id]
            own = torch.Generator(device=dev)
            own.manual_seed(int.from_bytes(os.urandom(7), "little") if seed is None else seed + 1000003 * (dev_id + 1))
            shared = torch.Generator(device=dev)
            shared.manual_seed(shared_seed)
            self._own[dev_id], self._shared[dev_id] = own, shared

    def _uniform_below(self, q, gen, dev):
        """One row of N integers uniform in [0, q)."""
        q = int(q)
        if q < (1 << 62):
            return torch.randint(0, q, (self.num_coefs,), generator=gen, device=dev, dtype=torch.int64)
        raise ValueError("modulus too large")

    def randint(self, amax=3, shift=0, repeats=0):
        """amax scalar: [1 - repeats own rows + repeats shared rows, N] in [shift, amax + shift);
        amax per-device lists of per-row moduli: one row per modulus, the last `repeats` rows shared."""
        if not isinstance(amax, (list, tuple)):
            amax = [[amax] for _ in self.shares]
        out = []
        for dev_id, moduli in enumerate(amax):
            if dev_id not in self._own:
                out.append(None)
                continue
            dev = self.devices[dev_id]
            moduli = list(moduli)
            n_own = len(moduli) - repeats
            rows = [self._uniform_below(q, self._own[dev_id], dev) for q in moduli[:n_own]]
            rows += [self._uniform_below(q, self._shared[dev_id], dev) for q in moduli[n_own:]]
            t = torch.stack(rows) if rows else torch.empty((0, self.num_coefs), dtype=torch.int64, device=dev)
            out.append(t + shift if shift else t)
        return out

    def discrete_gaussian(self, non_repeats=0, repeats=1):
        """Rounded Gaussian, sigma = self.sigma, [non_repeats own + repeats shared rows, N] per device."""
        if not isinstance(non_repeats, (list, tuple)):
            non_repeats = [non_repeats] * self.num_devices
        out = []
        for dev_id, n_own in enumerate(non_repeats):
            if dev_id not in self._own:
                out.append(None)
                continue
            dev = self.devices[dev_id]
            parts = []
            for count, gen in ((n_own, self._own[dev_id]), (repeats, self._shared[dev_id])):
                if count:
                    x = torch.empty((count, self.num_coefs), dtype=torch.float64, device=dev)
                    x.normal_(0.0, self.sigma, generator=gen)
                    parts.append(torch.round(x).to(torch.int64))
            out.append(torch.cat(parts) if parts else torch.empty((0, self.num_coefs), dtype=torch.int64, device=dev))
        return out

    def randround(self, coef):
        """Stochastic rounding of an fp64 tensor on the first device: floor(x) + Bernoulli(frac(x))."""
        dev_id = self.local_ids[0]
        gen = self._shared[dev_id]
        u = torch.rand(coef.shape, generator=gen, device=coef.device, dtype=torch.float64)
        fl = torch.floor(coef)
        return (fl + (u < (coef - fl))).to(torch.int64)
