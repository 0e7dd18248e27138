"""Limb-sharded HIP engine with one process per rank, all ranks on the box's single GPU.  RCCL refuses two ranks on
one device, so gloo carries the messages: the product's point-to-point batches are built and cached against the DEVICE
buffers exactly as for RCCL and travel through the host-staging transport of tests/gloo_device_p2p.py.  Rehearses the
device / stream / shard plumbing of the real multi-GPU path with the real kernels; results must match the reference's
2-device digests, and one process driving the same number of logical devices."""
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
    from tests import gloo_device_p2p
    p2p_log = gloo_device_p2p.install()
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    eng = ckks_engine(devices=["cuda:0"], comm=DistComm(local_device="cuda:0"), **PARAMS)
    assert eng.local_ids == [rank] and eng.backend.name.startswith("hip")
    for name, ct in _ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            arr = shards[0].cpu().numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.npy"), arr)
    # every message of the product's batches was a device buffer (the branch an RCCL run takes), none a collective
    assert p2p_log and all(is_cuda for batch in p2p_log for _, _, _, is_cuda in batch)
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


def _gold_worker(rank, world, port, outdir, exchange="p2p"):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tests import gloo_device_p2p
    gloo_device_p2p.install()
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0"], comm=DistComm(local_device="cuda:0", exchange=exchange), **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    # eager launches, then the same ops with this rank's fixed-address launches replayed from HIP graphs (the default): the
    # first graphed call captures, the second and third replay
    for mode, graphs, reps in (("eager", False, 1), ("graph", True, 3)):
        eng.graph_sharded = graphs
        for _ in range(reps):
            res = (("cc_mult", eng.cc_mult(a, b, evk)), ("rotate", eng.rotate_single(a, rotk)))
        for name, ct in res:
            for comp, shards in enumerate(ct.data):
                np.save(os.path.join(outdir, f"{mode}.{name}.{comp}.{rank}.npy"), shards[0].cpu().numpy())
    assert any(k[0] == "sgraph" for k in eng._tables if isinstance(k, tuple)), "the graphed path was not taken"
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["p2p", "allgather"])
def test_two_ranks_gold_fused_exchange_equals_one_process_two_devices(exchange):
    """BASELINE configs[3] in miniature: gold limb-sharded over 2 ranks.  The digits travel group by group and
    each group is extended + transformed as it lands (lf_ks_fwd), the tail (lf_ks_tail) runs once — against one
    process driving two logical devices, which runs the undivided lf_ks_core on the same shards.  Both forms of the digit
    exchange (point-to-point batch; one padded all-gather on device buffers), eager launches and graph replay."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    port = 30700 + (os.getpid() % 1000) + (40 if exchange != "p2p" else 0)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_gold_worker, args=(2, port, outdir, exchange), nprocs=2, join=True)
        got = {f: np.load(os.path.join(outdir, f)) for f in os.listdir(outdir)}
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0", "cuda:0"], **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for name, ct in (("cc_mult", eng.cc_mult(a, b, evk)), ("rotate", eng.rotate_single(a, rotk))):
        for comp, shards in enumerate(ct.data):
            for rank, t in enumerate(shards):
                for mode in ("eager", "graph"):     # graph = third call: captured segments replayed around the live exchange
                    assert (got[f"{mode}.{name}.{comp}.{rank}.npy"] == t.cpu().numpy()).all(), (mode, name, comp, rank)


# ---- a level where a rank has run out of rows (the alive ranks must not wait for it) -----------------
def _deep_ops(eng, synth):
    """small ring over 3 devices: rows per device 2/2/2 at level 0, 2/2/1 at level 1, 2/2/- from level 2 on
    (rns_partition(6, 2, 3)): the rescale 1 -> 2 is fed by a rank that then holds nothing, the key switches at level 2
    and 3 run between ranks 0 and 1 only; two key switches back to back reuse the digit buffer."""
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    a1, b1 = synth.ciphertext(eng, 31, 1), synth.ciphertext(eng, 32, 1)
    prod = eng.cc_mult(a1, b1, evk)                       # level 1 -> 2
    rot = eng.rotate_single(prod, rotk)                   # level 2
    rot2 = eng.rotate_single(rot, rotk)                   # same buffers again
    deeper = eng.cc_mult(rot2, prod, evk)                 # level 2 -> 3
    return {"prod": prod, "rot": rot, "rot2": rot2, "deeper": deeper}


def _deep_worker(rank, world, port, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    from tests import gloo_device_p2p
    gloo_device_p2p.install()
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    eng = ckks_engine(devices=["cuda:0"], comm=DistComm(local_device="cuda:0"), **PARAMS)
    for name, ct in _deep_ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            arr = shards[0].cpu().numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name}.{comp}.{rank}.npy"), arr)
    dist.barrier()
    dist.destroy_process_group()


def test_three_ranks_with_a_rank_out_of_rows_equal_one_process_three_devices():
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    port = 31700 + (os.getpid() % 1000)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_deep_worker, args=(3, port, outdir), nprocs=3, join=True)
        got = {f: np.load(os.path.join(outdir, f)) for f in os.listdir(outdir)}
    eng = ckks_engine(devices=["cuda:0"] * 3, **PARAMS)
    assert eng.len_devices[2] == 2 and eng.len_devices[1] == 3
    for name, ct in _deep_ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            for rank in range(3):
                want = shards[rank].cpu().numpy() if rank < len(shards) else np.zeros((0, eng.ctx.N), dtype=np.int64)
                assert (got[f"{name}.{comp}.{rank}.npy"] == want).all(), (name, comp, rank)


# ---- RCCL itself, at the world size a one-GPU lease allows --------------------------------------------------------------
def test_rccl_world_size_one_under_the_sharded_engine():
    """tools/rccl_world1.py in a FRESH process: init_process_group("nccl", world_size=1) on the leased GPU, then
      * all_gather_into_tensor, broadcast, a self-addressed batch_isend_irecv, the empty batches and the padded all-gather of
        fhe/comm.py at one rank;
      * stream ordering: a collective behind a slow producer reads the producer's words, a consumer behind work.wait() reads the
        collective's (default and side stream, async and blocking form, and a point-to-point batch);
      * gold cc_mult / rotate_single (levels 0 and 9) on the SHARDED code path — DistComm(solo_sharded=True), exchange "p2p"
        and "allgather", eager launches and HIP-graph replay (first op eager + capture, then replays around the live
        exchange) — every row equal to the unsharded engine's, with the collectives counted at the backend.
    Everything of the multi-GPU path except xGMI traffic itself (eng.py:778-810, 999-1011 are the reference's exchange sites)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py"), "--preset", "gold"], env=env,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, f"no JSON line (rc {r.returncode}): {r.stderr[-1500:]}"
    rep = json.loads(lines[-1])
    assert rep["backend"] == "nccl" and rep["world_size"] == 1, rep
    bad = {k: v for k, v in rep["checks"].items() if not v.get("ok")}
    bad.update({k: v for k, v in rep["engine"].items() if isinstance(v, dict) and not v.get("ok")})
    assert not bad and rep["ok"] and r.returncode == 0, json.dumps(bad)[:2000] + r.stderr[-800:]
    for mode in ("p2p", "allgather"):
        e = rep["engine"][f"sharded_path_{mode}"]
        assert e["eager_rows_differing"] == 0 and e["graph_rows_differing"] == 0 and e["graph_segment_sets"] >= 2, e
    assert rep["engine"]["sharded_path_allgather"]["backend_calls"]["all_gather_into_tensor"] >= 4
    assert set(rep["checks"]) >= {"communicator", "all_gather_into_tensor", "broadcast", "batch_isend_irecv_self_addressed",
                                  "comm_patterns_one_rank", "stream_ordering"}
