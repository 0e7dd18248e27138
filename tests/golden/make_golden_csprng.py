"""Generate tests/golden/csprng_golden.json from the REFERENCE's own Python code (build container only):

  * ChaCha20 blocks from liberate.csprng.chacha20_naive.chacha20 (torch, runs on CPU) on seeded states,
  * the CDT binary tree from liberate.csprng.discrete_gaussian_sampler.build_CDT_binary_search_tree.

    python tests/golden/make_golden_csprng.py

The fixture holds inputs and expected outputs only (numbers), no reference source text.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from tests.golden import refdriver as rd  # noqa: E402
from tests.csprng_streams import (STREAM_CONFIGS, STREAM_KEY, STREAM_NONCE, digest, randround_input,  # noqa: E402
                                  stream_calls)


def streams():
    out = {}
    for name, cfg in STREAM_CONFIGS.items():
        r = rd.reference_csprng(cfg["num_coefs"], cfg["num_channels"], cfg["num_repeating_channels"], cfg["n_dev"],
                                STREAM_KEY, STREAM_NONCE)
        calls = []
        for label, method, kw in stream_calls(cfg):
            res = getattr(r, method)(**kw)
            calls.append({"label": label, "sha256": digest(res), "head": [t.ravel()[:4].tolist() for t in res]})
        rr = r.randround(torch.from_numpy(randround_input(cfg["num_coefs"])))
        calls.append({"label": "randround", "sha256": digest([rr]), "head": [rr[:12].tolist()]})
        calls.append({"label": "final_states", "sha256": digest(r.states), "head": [s[0, 12:14].tolist() for s in r.states]})
        out[name] = {"config": cfg, "calls": calls}
    return out


def main():
    # The reference package imports its CUDA extensions on import; refdriver installs stand-ins for
    # them (none is called below: both functions used here are the reference's pure-Python code).
    rd.load_reference()
    from liberate.csprng import chacha20_naive as naive
    from liberate.csprng import discrete_gaussian_sampler as dgs

    rng = np.random.Generator(np.random.PCG64(20250523))
    n = 24
    states = rng.integers(0, 1 << 32, size=(n, 16), dtype=np.int64)
    states[0, 12:14] = (0xFFFFFFFF, 0xFFFFFFFF)          # counter at the wrap
    states[1, :] = 0
    states[2, :] = 0xFFFFFFFF
    blocks = naive.chacha20(torch.from_numpy(states.T.copy())).numpy().T       # reference layout is [16, n]
    stepped = torch.from_numpy(states.T.copy())
    naive.increment_counter(stepped, 3 * (1 << 31) + 12345)
    btree, _ptr, size, depth = dgs.build_CDT_binary_search_tree(security_bits=128, sigma=3.2)
    out = {
        "generator": "tests/golden/make_golden_csprng.py",
        "chacha20": {"states": states.tolist(), "blocks": blocks.tolist(),
                     "step": 3 * (1 << 31) + 12345, "stepped": stepped.numpy().T.tolist()},
        "cdt": {"sigma": "3.2", "security_bits": 128, "size": int(size), "depth": int(depth),
                "btree_low_high": [[str(int(lo)), str(int(hi))] for lo, hi in btree]},
        "streams": {"key": STREAM_KEY, "nonce": STREAM_NONCE, "configs": streams()},
    }
    with open(os.path.join(HERE, "csprng_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote csprng_golden.json:", n, "blocks,", size, "tree nodes, depth", depth)


if __name__ == "__main__":
    main()
