"""Limb-sharded HIP engine with one process per rank, both ranks on the box's single GPU (gloo carries the
collectives; RCCL refuses two ranks on one device).  Rehearses the device / stream / shard plumbing of the
real multi-GPU path with the real kernels; results must match the reference's 2-device digests."""
import json
import os
import sys
import tempfile
import warnings

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.test_distributed_cpu import GOLD, PARAMS, _ops, _digest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    eng = ckks_engine(devices=["cuda:0"], comm=DistComm(local_device="cuda:0"), **PARAMS)
    assert eng.local_ids == [rank] and eng.backend.name.startswith("hip")
    for name, ct in _ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            arr = shards[0].cpu().numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.npy"), arr)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_reference_digests():
    port = 29700 + (os.getpid() % 1000)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(2, port, outdir), nprocs=2, join=True)
        got = {}
        for f in os.listdir(outdir):
            name, comp, rank, _ = f.rsplit(".", 3)
            got.setdefault(name, {}).setdefault(int(comp), {})[int(rank)] = np.load(os.path.join(outdir, f))
    want = GOLD["small_x2"]["ops"]
    for name, comps in want.items():
        for comp, rec in enumerate(comps):
            assert _digest(got[name][comp]) == rec["sha256"], (name, comp)


def _gold_worker(rank, world, port, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0"], comm=DistComm(local_device="cuda:0"), **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for name, ct in (("cc_mult", eng.cc_mult(a, b, evk)), ("rotate", eng.rotate_single(a, rotk))):
        for comp, shards in enumerate(ct.data):
            np.save(os.path.join(outdir, f"{name}.{comp}.{rank}.npy"), shards[0].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gold_fused_exchange_equals_one_process_two_devices():
    """BASELINE configs[3] in miniature: gold limb-sharded over 2 ranks.  The digits travel group by group and
    each group is extended + transformed as it lands (lf_ks_fwd), the tail (lf_ks_tail) runs once — against one
    process driving two logical devices, which runs the undivided lf_ks_core on the same shards."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    port = 30700 + (os.getpid() % 1000)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_gold_worker, args=(2, port, outdir), nprocs=2, join=True)
        got = {f: np.load(os.path.join(outdir, f)) for f in os.listdir(outdir)}
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0", "cuda:0"], **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for name, ct in (("cc_mult", eng.cc_mult(a, b, evk)), ("rotate", eng.rotate_single(a, rotk))):
        for comp, shards in enumerate(ct.data):
            for rank, t in enumerate(shards):
                assert (got[f"{name}.{comp}.{rank}.npy"] == t.cpu().numpy()).all(), (name, comp, rank)
