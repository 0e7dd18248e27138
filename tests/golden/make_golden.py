"""Generate tests/golden/engine_digests.json by running the REFERENCE engine (imported from
/root/reference, CUDA modules replaced by the C oracle — see refdriver.py) on synthetic inputs.

    python tests/golden/make_golden.py            # build container only

Inputs are regenerated from seeds by liberate_fhe_amd.utils.synth (splitmix64 -> mod q), so the GPU
box reproduces them without the reference; the fixture stores only SHA-256 digests and a few sample
words of the expected outputs.  No reference source text is stored, only inputs' seeds and outputs.
"""
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from tests.golden import refdriver as rd  # noqa: E402
from liberate_fhe_amd.utils import synth  # noqa: E402

CONFIGS = {
    "small": dict(logN=12, num_scales=5, num_special_primes=2, is_secured=False),
    "bronze": dict(logN=14, num_special_primes=1),
    "silver": dict(logN=15, num_special_primes=2),
    "gold": dict(logN=16, num_special_primes=4),
}


def digest(ct):
    """One digest per component over all devices' rows re-ordered to natural prime order."""
    out = []
    for comp in ct.data:
        h = hashlib.sha256()
        for t in comp:
            h.update(np.ascontiguousarray(t.numpy()).tobytes())
        first = comp[0]
        out.append({"sha256": h.hexdigest(), "shape": [list(t.shape) for t in comp],
                    "head": [int(x) for x in first[0, :4]], "tail": [int(x) for x in first[-1, -4:]]})
    return out


def run(name, params, n_dev=1):
    t0 = time.time()
    eng = rd.reference_engine(n_dev, **params)
    rec = {"params": params, "n_devices": n_dev, "q": [int(x) for x in eng.ctx.q], "hash": eng.hash, "ops": {}}
    a, b = synth.ciphertext(eng, 11, 0), synth.ciphertext(eng, 12, 0)
    evk = synth.key_switch_key(eng, 13)
    rotk = synth.key_switch_key(eng, 14, origin="rotation key:7")
    rec["seeds"] = {"ct_a": 11, "ct_b": 12, "evk": 13, "rotk": 14, "rot_delta": 7}
    rec["ops"]["rescale(a)"] = digest(eng.rescale(a))
    prod = eng.cc_mult(a, b, evk)
    rec["ops"]["cc_mult(a,b,evk)"] = digest(prod)
    rec["ops"]["rotate_single(a,rotk)"] = digest(eng.rotate_single(a, rotk))
    rec["ops"]["rotate_single(cc_mult,rotk)"] = digest(eng.rotate_single(prod, rotk))
    rec["ops"]["cc_add(a,b)"] = digest(eng.cc_add(a, b))
    if name != "gold":
        rec["ops"]["cc_mult(prod,prod,evk)"] = digest(eng.cc_mult(prod, prod, evk))
    print(f"{name} x{n_dev}: {time.time() - t0:.1f} s", flush=True)
    return rec


if __name__ == "__main__":
    which = sys.argv[1:] or list(CONFIGS)
    path = os.path.join(HERE, "engine_digests.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name in which:
        out[name] = run(name, CONFIGS[name])
        if name == "small":
            out["small_x2"] = run(name, CONFIGS[name], 2)
        json.dump(out, open(path, "w"), indent=1)
