"""Limb -> (GPU, key-switch digit) map of the RNS basis, per level.

Host-side mirror of the reference's `rns_partition` (src/liberate/ntt/rns_partition.py:4-170); the
attribute names and nesting are the reference's because the engine-level code indexes them directly
(`p.destination_arrays[level][dev]`, `p.p[level][dev][part]`, `p.rescaler_loc[level]`, ...).

Layout rules being reproduced:
  * limbs 0..L-2 are scale primes, L-1 the base prime, L..L+K-1 the special primes;
  * key-switch digits ("partitions") are runs of K consecutive scale primes, then [base], then the
    specials as one group (part.py:8-27);
  * digit j goes to GPU (P-1-j) mod D — walking down from the last digit so GPU 0 holds the highest
    one — GPU 0 additionally owns the base-prime digit, and EVERY GPU carries the special limbs
    (part.py:29-46), which is what lets the mod-down after key-switching run without an exchange;
  * at level l the limbs < l are gone; row indices are re-based per GPU.
"""
from __future__ import annotations

import numpy as np


class rns_partition:
    def __init__(self, num_ordinary_primes=17, num_special_primes=2, num_devices=2, balance=False):
        """balance (NOT the reference's layout; default off): GPU 0 carries the base-prime digit on top of its share of the
        scale-prime digits, and in the reference's walk it is also the GPU that receives the (short) top digit — gold over 8
        GPUs: 2 + 4 + 1 = 7 rows against 4 on the other seven, so every limb-sharded op waits for rank 0 (37 % more transform
        work, 3.67 MB per link against 2.1).  With balance=True the top digit goes to GPU 1 when that lowers the largest row
        count (gold / 8: 5, 6, 4, 4, 4, 4, 4, 4).  Digits are unchanged — same key-switch decomposition, same canonical words
        per limb — and GPUs still run out of rows from the last one down (GPU 0 keeps the base prime, GPU 1 the highest scale
        primes), which is what the engine's per-level device lists assume.  Digit sizes cap what is reachable: eight digits of
        4, one of 2 and one of 1 cannot be spread flatter than 6 / 5 / 4 x 6 without re-cutting digits, i.e. another key."""
        L, K, D = num_ordinary_primes, num_special_primes, num_devices
        self.balance = bool(balance)
        self.num_ordinary_primes, self.num_special_primes, self.num_devices = L, K, D
        self.num_scales = L - 1
        self.base_prime_idx = L - 1

        P = -(-(L - 1) // K)  # number of scale-prime digits
        self.num_partitions = P
        digits = [list(range(j * K, min((j + 1) * K, L - 1))) for j in range(P)]
        self.partitions = digits + [[L - 1]] + [list(range(L, L + K))]

        shares = [sorted(range(P - 1 - dev, -1, -D)) for dev in range(D)]
        if self.balance and D >= 2 and len(shares[0]) >= 1 and shares[0][-1] == P - 1:
            rows = lambda share, base: sum(len(digits[j]) for j in share) + base
            before = max(rows(shares[dev], 1 if dev == 0 else 0) for dev in range(D))
            moved = [shares[0][:-1], sorted(shares[1] + [P - 1])] + shares[2:]
            after = max(rows(moved[dev], 1 if dev == 0 else 0) for dev in range(D))
            if after < before:
                shares = moved
        self.part_allocations = []
        for dev in range(D):
            mine = list(shares[dev])
            if dev == 0:
                mine.append(P)
            mine.append(P + 1)
            self.part_allocations.append(mine)
        self.prime_allocations = [[self.partitions[j] for j in alloc] for alloc in self.part_allocations]
        self.flat_prime_allocations = [[x for part in alloc for x in part] for alloc in self.prime_allocations]

        self.compute_destination_arrays()
        self.compute_rescaler_locations()
        self.compute_partitions()

    # rows alive at each level, per GPU (part.py:64-84)
    def compute_destination_arrays(self):
        L, K = self.num_ordinary_primes, self.num_special_primes
        self.destination_arrays_with_special = [
            [[x for x in flat if x >= lvl] for flat in self.flat_prime_allocations] for lvl in range(L)
        ]
        # ordinary-only lists drop GPUs that have run out of ordinary limbs
        self.destination_arrays = [
            [rows[:-K] for rows in per_dev if len(rows) > K] for per_dev in self.destination_arrays_with_special
        ]

    # the GPU holding the next limb to be dropped (part.py:86-91)
    def compute_rescaler_locations(self):
        self.rescaler_loc = []
        for per_dev in self.destination_arrays_with_special:
            lows = [min(rows) for rows in per_dev]
            self.rescaler_loc.append(lows.index(min(lows)))

    def partings(self, lvl):
        """Per GPU at level `lvl`: cumulative digit ends, digit sizes and digit row-index lists."""
        cums, counts, parts = [], [], []
        for dev in range(self.num_devices):
            gone = len(self.destination_arrays_with_special[0][dev]) - len(self.destination_arrays_with_special[lvl][dev])
            ends = np.cumsum([len(part) for part in self.prime_allocations[dev]]) - gone
            ends = [int(e) for e in ends if e > 0]
            cums.append(ends)
            counts.append(np.diff(np.array(ends), prepend=0))
            parts.append([list(range(a, b)) for a, b in zip([0] + ends[:-1], ends)])
        return cums, counts, parts

    # per level: digit row ranges in level-local (`parts`) and level-0 (`p`) row numbering (part.py:93-170)
    def compute_partitions(self):
        D = self.num_devices
        self.part_cumsums, self.part_counts, self.parts = [], [], []
        self.destination_parts, self.destination_parts_with_special = [], []
        self.p, self.p_special, self.diff = [], [], []
        self.d = [self.destination_arrays[0][dev] for dev in range(D)]
        self.d_special = [self.destination_arrays_with_special[0][dev] for dev in range(D)]

        for lvl in range(self.num_ordinary_primes):
            cums, counts, parts = self.partings(lvl)
            self.part_cumsums.append(cums)
            self.part_counts.append(counts)
            self.parts.append(parts)

            rows_now = self.destination_arrays_with_special[lvl]
            with_special = [[[rows[i] for i in part] for part in dev_parts] for rows, dev_parts in zip(rows_now, parts)]
            self.destination_parts_with_special.append(with_special)
            self.destination_parts.append([dev_parts[:-1] for dev_parts in with_special])

            gone = [len(r0) - len(r1) for r0, r1 in zip(self.destination_arrays_with_special[0], rows_now)]
            shifted = [[[i + g for i in part] for part in dev_parts] for g, dev_parts in zip(gone, parts)]
            self.p_special.append(shifted)
            self.p.append([dev_parts[:-1] for dev_parts in shifted])
            self.diff.append(gone)
