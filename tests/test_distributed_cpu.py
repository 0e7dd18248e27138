"""Limb-sharded engine with one process per rank (world_size 2 and 3, gloo on CPU).

Each rank owns the limbs `rns_partition` gives its device, the rescale row travels by broadcast and the
key-switch digits by all-gather (liberate_fhe_amd/fhe/comm.py); arithmetic is the checker backend so
the test needs no GPU.  The combined shards must reproduce the golden digests that the REFERENCE engine
produced with the same number of in-process devices (tests/golden/engine_digests.json, "small_x2"),
and must equal this repo's single-process multi-device run for world_size 3.
"""
import hashlib
import json
import os
import sys
import tempfile
import warnings

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "engine_digests.json")))
PARAMS = GOLD["small"]["params"]


def _ops(eng, synth):
    s = GOLD["small"]["seeds"]
    a, b = synth.ciphertext(eng, s["ct_a"], 0), synth.ciphertext(eng, s["ct_b"], 0)
    evk = synth.key_switch_key(eng, s["evk"])
    rotk = synth.key_switch_key(eng, s["rotk"], origin=f"rotation key:{s['rot_delta']}")
    prod = eng.cc_mult(a, b, evk)
    return {"rescale(a)": eng.rescale(a), "cc_mult(a,b,evk)": prod, "rotate_single(a,rotk)": eng.rotate_single(a, rotk),
            "rotate_single(cc_mult,rotk)": eng.rotate_single(prod, rotk), "cc_add(a,b)": eng.cc_add(a, b),
            "cc_mult(prod,prod,evk)": eng.cc_mult(prod, prod, evk)}


def _worker(rank, world, port, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), comm=DistComm(local_device="cpu"), **PARAMS)
    assert eng.local_ids == [rank]
    for name, ct in _ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            assert len(shards) <= 1
            arr = shards[0].numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.npy"), arr)
    dist.barrier()
    dist.destroy_process_group()


def _run(world):
    port = 29500 + (os.getpid() % 2000) + world
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, port, outdir), nprocs=world, join=True)
        out = {}
        for f in os.listdir(outdir):
            name, comp, rank, _ = f.rsplit(".", 3)
            out.setdefault(name, {}).setdefault(int(comp), {})[int(rank)] = np.load(os.path.join(outdir, f))
        return out


def _digest(shards_by_rank):
    h = hashlib.sha256()
    for r in sorted(shards_by_rank):
        if shards_by_rank[r].shape[0]:
            h.update(np.ascontiguousarray(shards_by_rank[r]).tobytes())
    return h.hexdigest()


def test_two_ranks_reproduce_reference_two_device_digests():
    got = _run(2)
    want = GOLD["small_x2"]["ops"]
    assert set(got) == set(want)
    for name, comps in want.items():
        for comp, rec in enumerate(comps):
            assert _digest(got[name][comp]) == rec["sha256"], (name, comp)


def test_three_ranks_equal_single_process_three_devices():
    warnings.filterwarnings("ignore")
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"] * 3, backend=OracleBackend(), **PARAMS)
    want = _ops(eng, synth)
    got = _run(3)
    for name, ct in want.items():
        for comp, tensors in enumerate(ct.data):
            h = hashlib.sha256()
            for t in tensors:
                h.update(np.ascontiguousarray(t.numpy()).tobytes())
            assert _digest(got[name][comp]) == h.hexdigest(), (name, comp)
