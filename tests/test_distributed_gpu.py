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
