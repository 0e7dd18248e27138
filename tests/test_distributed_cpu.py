"""Limb-sharded engine with one process per rank (world_size 2 and 3, gloo on CPU).

Each rank owns the limbs `rns_partition` gives its device, the rescale row travels by broadcast and the
key-switch digits by one in-place broadcast per run of same-owner digits (liberate_fhe_amd/fhe/comm.py); arithmetic is the checker backend so
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


# ---- the exchange schedule of the fused key switch (logN >= 13) ---------------------------------------------------
FUSED = dict(logN=13, num_scales=6, num_special_primes=2, is_secured=False)


def _schedule_worker(rank, world, port, outdir):
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
    log = []

    class SpyWork:
        def __init__(self, work, tag):
            self.work, self.tag = work, tag

        def wait(self):
            log.append(("wait", self.tag))
            return self.work.wait()

    class SpyComm(DistComm):
        def broadcast_into(self, buf, src, async_op=False):
            log.append(("bcast", src, buf.data_ptr(), tuple(buf.shape), async_op))
            w = super().broadcast_into(buf, src, async_op=async_op)
            return SpyWork(w, (src, buf.data_ptr())) if async_op else w

        def all_gather(self, tensor):
            log.append(("all_gather",))
            return super().all_gather(tensor)

    class SpyBackend(OracleBackend):
        def ks_fwd(self, state, first, count, *a, **k):
            log.append(("ks_fwd", first, count))
            return super().ks_fwd(state, first, count, *a, **k)

        def ks_tail(self, *a, **k):
            log.append(("ks_tail",))
            return super().ks_tail(*a, **k)

    eng = ckks_engine(devices=["cpu"], backend=SpyBackend(), comm=SpyComm(local_device="cpu"), **FUSED)
    a = synth.ciphertext(eng, 3, 0)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    runs = []
    for _ in range(2):
        del log[:]
        out = eng.rotate_single(a, rotk)
        runs.append(list(log))
    tabs = eng._ks_tables(0)
    np.save(os.path.join(outdir, f"out.{rank}.npy"), np.stack([out.data[0][0].numpy(), out.data[1][0].numpy()]))
    import pickle
    pickle.dump({"runs": runs, "groups": tabs["groups"], "total_rows": tabs["total_rows"], "N": eng.ctx.N},
                open(os.path.join(outdir, f"log.{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_key_switch_exchanges_per_digit_group_into_preallocated_buffers():
    """One in-place broadcast per run of same-owner digits, un-padded, all issued before the first wait; each
    group's extension + NTT (ks_fwd) starts right after ITS wait, the tail after the last; no all-gather, no
    per-call buffer (the same addresses in a second call); results equal the single-process two-device run."""
    import pickle
    warnings.filterwarnings("ignore")
    world = 2
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_schedule_worker, args=(world, port, outdir), nprocs=world, join=True)
        logs = [pickle.load(open(os.path.join(outdir, f"log.{r}.pkl"), "rb")) for r in range(world)]
        outs = [np.load(os.path.join(outdir, f"out.{r}.npy")) for r in range(world)]
    for rec in logs:
        groups, N = rec["groups"], rec["N"]
        assert len(groups) >= 2
        first, second = rec["runs"]
        for run in (first, second):
            assert ("all_gather",) not in run
            bc = [e for e in run if e[0] == "bcast"]
            assert [e[1] for e in bc] == [g[0] for g in groups]                       # one message per group, from its owner
            assert [e[3] for e in bc] == [(g[4], N) for g in groups]                  # exactly the group's rows: no padding
            assert all(e[4] for e in bc)                                              # asynchronous
            # every message is issued before anything is waited for
            assert max(i for i, e in enumerate(run) if e[0] == "bcast") < min(i for i, e in enumerate(run) if e[0] == "wait")
            # wait(g) -> ks_fwd(g) -> wait(g+1) -> ... -> ks_tail
            tail = [e for e in run if e[0] in ("wait", "ks_fwd", "ks_tail")]
            want = []
            for g, e in zip(groups, bc):
                want += [("wait", (g[0], e[2])), ("ks_fwd", g[1], g[2])]
            assert tail == want + [("ks_tail",)]
        # destination rows of one storage-order buffer, the same allocation in both calls
        p1 = [e[2] for e in first if e[0] == "bcast"]
        p2 = [e[2] for e in second if e[0] == "bcast"]
        assert p1 == p2
        assert [p - p1[0] for p in p1] == [(g[3] - groups[0][3]) * N * 8 for g in groups]
    # same words as one process driving two devices
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"] * 2, backend=OracleBackend(), **FUSED)
    want = eng.rotate_single(synth.ciphertext(eng, 3, 0), synth.key_switch_key(eng, 6, origin="rotation key:1"))
    for r in range(world):
        assert (outs[r][0] == want.data[0][r].numpy()).all() and (outs[r][1] == want.data[1][r].numpy()).all()
