"""ChaCha20-based samplers behind keygen / encrypt / encode: the reference's `Csprng`
(src/liberate/csprng/csprng.py:18-323) on the HIP kernels of csrc/ckks_csprng.hip.

State layout, counter assignment and stream consumption follow the reference:
  * one ChaCha20 state (16 int64 words holding 32-bit values) per 4 coefficients, L = N/4 states per
    RNS channel, `shares[d]` own channels on logical GPU d followed by `num_repeating_channels`
    channels whose states are identical on every GPU (csprng.py:80-160);
  * word 12/13 = block counter: own channels number their states globally across GPUs, the repeating
    channels continue after the last own channel; every draw advances the drawn states by
    inc = (sum(shares) + num_repeating_channels) * L (csprng.py:96-112, 145-158);
  * words 0-3 "expand 32-byte k", 4-11 the 256-bit key (same key on every GPU), 14-15 the nonce
    (csprng.py:170-190).

One process per GPU (`local_ids` = the logical GPUs this process owns): only the local state tables
exist, counters are computed from the global layout, so the union over ranks is exactly the
single-process stream, provided every rank passes the same `seed` / `nonce` (the engine broadcasts
them from rank 0).  Output lists keep one slot per logical GPU, None for GPUs owned by other ranks.

Differences from the reference, all deliberate:
  * own-channel counters are cumulative over GPUs (GPU d starts after all channels of GPUs < d).  The
    reference starts GPU d >= 1 at shares[d-1] * L (csprng.py:96: `[0] + [s * L for s in shares[:-1]]`,
    not a running sum), so with three or more GPUs the GPUs 1, 2, ... draw from the SAME ChaCha20 blocks
    and their "independent" uniform limbs are functions of one another.  `reference_counter_layout=True`
    reproduces the reference's numbering bit for bit (used by the parity tests); the default does not.
  * `seed` (8 x 32-bit words) and `nonce` (2 words) are honoured when given; the reference accepts
    the arguments but always draws from os.urandom (csprng.py:216-223).
  * `randround` on a multi-process engine draws from the first repeating channel, so every rank rounds
    a replicated plaintext identically; single-process it uses the first N/16 states of GPU 0 like
    the reference (csprng.py:312-321).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import chacha20_cuda, discrete_gaussian_cuda, randint_cuda, randround_cuda
from .discrete_gaussian_sampler import build_CDT_binary_search_tree

SIGMA_WORDS = tuple(int.from_bytes(w, "little") for w in (b"expa", b"nd 3", b"2-by", b"te k"))


def _words(value, count, what):
    """`count` 32-bit words: from a list, or fresh from the OS."""
    if value is None:
        return [int.from_bytes(os.urandom(4), "big") for _ in range(count)]
    words = [int(w) for w in value]
    if len(words) != count or any(not 0 <= w < (1 << 32) for w in words):
        raise ValueError(f"{what} must be {count} words of 32 bits")
    return words


class Csprng:
    # The four kernel modules (same names as the reference's extensions).  The CPU test-suite swaps in
    # its checker's modules here to exercise the host logic without a GPU; the product never does.
    chacha20_cuda = chacha20_cuda
    randint_cuda = randint_cuda
    discrete_gaussian_cuda = discrete_gaussian_cuda
    randround_cuda = randround_cuda

    def __init__(self, num_coefs=2 ** 15, num_channels=(8,), num_repeating_channels=2, sigma=3.2,
                 devices=None, seed=None, nonce=None, local_ids=None,
                 reference_counter_layout=False):
        self.num_coefs = num_coefs
        self.num_channels = list(num_channels)
        self.num_repeating_channels = num_repeating_channels
        self.sigma = sigma
        if devices is None:
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        self.devices = list(devices)
        self.num_devices = len(self.devices)
        self.local_ids = list(range(self.num_devices)) if local_ids is None else list(local_ids)
        if len(self.num_channels) == 1:
            self.shares = self.num_channels * self.num_devices
        elif len(self.num_channels) == self.num_devices:
            self.shares = self.num_channels
        else:
            raise Exception("There was a contradicting mismatch between num_channels, and devices.")
        self.total_num_channels = sum(self.shares)
        if num_coefs % 16:
            raise ValueError("num_coefs must be a multiple of 16")
        self.L = num_coefs // 4

        self.btree, self.btree_ptr, self.btree_size, self.tree_depth = build_CDT_binary_search_tree(
            security_bits=128, sigma=sigma)

        # Global counter layout.
        if reference_counter_layout:
            self.start_ind = [0] + [s * self.L for s in self.shares[:-1]]
        else:
            self.start_ind = [0]
            for s in self.shares[:-1]:
                self.start_ind.append(self.start_ind[-1] + s * self.L)
        self.inc = (self.total_num_channels + num_repeating_channels) * self.L
        self.repeating_start = self.total_num_channels * self.L

        self.states = [None] * self.num_devices
        self.channeled_states = [None] * self.num_devices
        self.counters = [None] * self.num_devices
        for d in self.local_ids:
            rows = (self.shares[d] + num_repeating_channels) * self.L
            self.states[d] = torch.zeros((rows, 16), dtype=torch.int64, device=self.devices[d])
            self.channeled_states[d] = self.states[d].view(self.shares[d] + num_repeating_channels, self.L, 16)
            own = torch.arange(self.start_ind[d], self.start_ind[d] + self.shares[d] * self.L, dtype=torch.int64)
            rep = torch.arange(self.repeating_start, self.inc, dtype=torch.int64)
            self.counters[d] = torch.cat([own, rep]).to(self.devices[d])
        self.refresh(seed, nonce)

    # -- state ------------------------------------------------------------------------------------

    def refresh(self, seed=None, nonce=None):
        key = _words(seed, 8, "seed")
        non = _words(nonce, 2, "nonce")
        self.key = [None] * self.num_devices
        self.nonce = [None] * self.num_devices
        for d in self.local_ids:
            dev = self.devices[d]
            self.key[d] = torch.tensor(key, dtype=torch.int64, device=dev)
            self.nonce[d] = torch.tensor(non, dtype=torch.int64, device=dev)
            self.initialize_states(d)

    def initialize_states(self, dev_id, seed=None, nonce=None):
        state = self.states[dev_id]
        state.zero_()
        state[:, 0:4] = torch.tensor(SIGMA_WORDS, dtype=torch.int64, device=state.device)[None, :]
        state[:, 4:12] = self.key[dev_id][None, :]
        state[:, 12] = self.counters[dev_id]        # low counter word; the kernels carry into word 13
        state[:, 14:16] = self.nonce[dev_id][None, :]

    def _slice(self, dev_id, own, repeats):
        """The last `own` own channels and the first `repeats` repeating channels of one GPU."""
        if own > self.shares[dev_id] or repeats > self.num_repeating_channels or own < 0:
            raise ValueError("more channels requested than were procured")
        start = self.shares[dev_id] - own
        return self.channeled_states[dev_id][start:self.shares[dev_id] + repeats]

    def _scatter(self, results):
        out = [None] * self.num_devices
        for d, r in zip(self.local_ids, results):
            out[d] = r
        return out

    # -- samplers ---------------------------------------------------------------------------------

    def randbytes(self, shares=None, repeats=0, reshape=False):
        """(shares_d + repeats) * L random 64-byte blocks per GPU, as [.., 16] 32-bit words."""
        if shares is None:
            shares = self.shares
        targets = [self._slice(d, shares[d], repeats).view(-1, 16) for d in self.local_ids]
        blocks = self.chacha20_cuda.chacha20(targets, self.inc)
        if reshape:
            blocks = [b.view(-1, self.L, 16) for b in blocks]
        return self._scatter(blocks)

    def randint(self, amax=3, shift=0, repeats=0):
        """amax scalar: one channel per GPU in [shift, amax + shift) (the repeating one when repeats = 1);
        amax = per-GPU lists of per-channel moduli: one row per modulus, the trailing `repeats` rows
        identical on every GPU."""
        if not isinstance(amax, (list, tuple)):
            amax = [[amax] for _ in self.shares]
        targets, q_host = [], []
        for d in self.local_ids:
            moduli = np.ascontiguousarray([int(q) for q in amax[d]], dtype=np.uint64)
            targets.append(self._slice(d, len(moduli) - repeats, repeats))
            q_host.append(moduli)
        return self._scatter(self.randint_cuda.randint_fast(targets, q_host, shift, self.inc))

    def discrete_gaussian(self, non_repeats=0, repeats=1):
        """[non_repeats_d + repeats, N] samples of the discrete Gaussian (sigma = self.sigma) per GPU."""
        if not isinstance(non_repeats, (list, tuple)):
            non_repeats = [non_repeats] * self.num_devices
        targets = [self._slice(d, non_repeats[d], repeats).view(-1, 16) for d in self.local_ids]
        flat = self.discrete_gaussian_cuda.discrete_gaussian_fast(targets, self.btree_ptr, self.btree_size,
                                                             self.tree_depth, self.inc)
        return self._scatter([f.view(-1, self.num_coefs) for f in flat])

    def randround(self, coef):
        """Stochastic rounding of an fp64 vector of N coefficients on this process's first GPU."""
        d = self.local_ids[0]
        L = self.num_coefs // 16
        if len(self.local_ids) == self.num_devices:
            states = self.states[d][:L]
        else:
            states = self.channeled_states[d][self.shares[d]][:L]
        rand_bytes = self.chacha20_cuda.chacha20((states,), self.inc)[0].ravel()
        coef = coef.contiguous()
        if coef.numel() != rand_bytes.numel():
            raise ValueError(f"randround expects {self.num_coefs} coefficients")
        self.randround_cuda.randround([coef], [rand_bytes])
        return rand_bytes
