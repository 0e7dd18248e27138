"""Limb-sharded engine with one process per rank (world_size 2 and 3, gloo on CPU).

Each rank owns the limbs `rns_partition` gives its device, the rescale row travels by broadcast and the
key-switch digits by one batch of in-place point-to-point messages (liberate_fhe_amd/fhe/comm.py); arithmetic is the checker backend so
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


def _worker(rank, world, port, outdir, exchange="p2p", solo=False):
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
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(),
                      comm=DistComm(local_device="cpu", exchange=exchange, solo_sharded=solo), **PARAMS)
    assert eng.local_ids == [rank] and eng._multi == (world > 1 or solo)
    for name, ct in _ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            assert len(shards) <= 1
            arr = shards[0].numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.npy"), arr)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, exchange="p2p", solo=False):
    port = 29500 + (os.getpid() % 2000) + world + (50 if exchange != "p2p" else 0)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, port, outdir, exchange, solo), nprocs=world, join=True)
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


@pytest.mark.parametrize("exchange", ["p2p", "allgather"])
def test_two_ranks_reproduce_reference_two_device_digests(exchange):
    """Both forms of the digit exchange (point-to-point batch, padded all-gather) against the REFERENCE's digests."""
    got = _run(2, exchange)
    want = GOLD["small_x2"]["ops"]
    assert set(got) == set(want)
    for name, comps in want.items():
        for comp, rec in enumerate(comps):
            assert _digest(got[name][comp]) == rec["sha256"], (name, comp)


@pytest.mark.parametrize("exchange", ["p2p", "allgather"])
def test_one_rank_on_the_sharded_code_path_reproduces_single_device_digests(exchange):
    """DistComm(solo_sharded=True): a group of ONE rank keeps the limb-sharded code path and issues both exchange steps to the
    communicator (what tools/rccl_world1.py runs on a real RCCL communicator on the GPU box); the words are the one-device
    reference digests.  Without the flag a one-rank communicator is ignored (same digests, the ordinary path)."""
    want = GOLD["small"]["ops"]
    for solo in (True, False):
        got = _run(1, exchange, solo=solo)
        for name, comps in got.items():
            for comp in comps:
                assert _digest(got[name][comp]) == want[name][comp]["sha256"], (solo, name, comp)


@pytest.mark.parametrize("exchange", ["p2p", "allgather"])
def test_three_ranks_equal_single_process_three_devices(exchange):
    warnings.filterwarnings("ignore")
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"] * 3, backend=OracleBackend(), **PARAMS)
    want = _ops(eng, synth)
    got = _run(3, exchange)
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

    class SpyWorks:
        def __init__(self, works):
            self.works = works

        def wait(self):
            log.append(("wait",))
            return self.works.wait()

    class SpyComm(DistComm):
        def exchange_rows(self, buf, pieces, peers):
            log.append(("exchange", buf.data_ptr(), tuple(buf.shape), tuple(pieces), tuple(peers)))
            return SpyWorks(super().exchange_rows(buf, pieces, peers))

        def broadcast(self, *a, **k):
            log.append(("broadcast",))
            return super().broadcast(*a, **k)

        def all_gather(self, tensor):
            log.append(("all_gather",))
            return super().all_gather(tensor)

    real_batch = dist.batch_isend_irecv

    def spy_batch(ops):
        log.append(("p2p", tuple(("send" if op.op is dist.isend else "recv", op.peer, tuple(op.tensor.shape)) for op in ops)))
        return real_batch(ops)
    dist.batch_isend_irecv = spy_batch

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
    pickle.dump({"runs": runs, "groups": tabs["groups"], "total_rows": tabs["total_rows"], "N": eng.ctx.N, "rank": rank},
                open(os.path.join(outdir, f"log.{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_key_switch_exchanges_point_to_point_into_preallocated_buffers(world):
    """ONE batch of point-to-point messages per key switch: a rank sends each run of digits it owns to every other rank
    and receives every other run from its owner, un-padded, straight into its rows of the storage-order buffer (the same
    allocation in a second call); no collective.  The rank's own digits are extended + transformed (ks_fwd) BEFORE the
    single wait, the foreign runs after it, the tail last; results equal the single-process multi-device run."""
    import pickle
    warnings.filterwarnings("ignore")
    port = 31500 + (os.getpid() % 2000) + world
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_schedule_worker, args=(world, port, outdir), nprocs=world, join=True)
        logs = [pickle.load(open(os.path.join(outdir, f"log.{r}.pkl"), "rb")) for r in range(world)]
        outs = [np.load(os.path.join(outdir, f"out.{r}.npy")) for r in range(world)]
    for rec in logs:
        groups, N, me = rec["groups"], rec["N"], rec["rank"]
        assert len(groups) >= 2
        first, second = rec["runs"]
        for run in (first, second):
            assert ("all_gather",) not in run and ("broadcast",) not in run
            ex = [e for e in run if e[0] == "exchange"]
            assert len(ex) == 1 and ex[0][2] == (rec["total_rows"], N)
            assert ex[0][3] == tuple((g[0], g[3], g[4]) for g in groups) and ex[0][4] == tuple(range(world))
            p2p = [e for e in run if e[0] == "p2p"]
            assert len(p2p) == 1
            want_ops = []
            for g in groups:
                if g[0] == me:
                    want_ops += [("send", p, (g[4], N)) for p in range(world) if p != me]
                else:
                    want_ops.append(("recv", g[0], (g[4], N)))
            assert list(p2p[0][1]) == want_ops                                        # exactly the rows, no padding
            # own digits before the wait, every foreign digit after it, in runs; then the tail
            seq = [e for e in run if e[0] in ("wait", "ks_fwd", "ks_tail")]
            assert seq.count(("wait",)) == 1 and seq[-1] == ("ks_tail",)
            w = seq.index(("wait",))
            own = [(g[1], g[2]) for g in groups if g[0] == me]
            assert [(e[1], e[2]) for e in seq[:w]] == own
            after = [(e[1], e[2]) for e in seq[w + 1:-1]]
            covered = sorted(d for f, c in own + after for d in range(f, f + c))
            assert covered == list(range(sum(g[2] for g in groups)))                  # every digit exactly once
            assert all(g[0] != me for g in groups for f, c in after if f <= g[1] < f + c)
        assert first[[e[0] for e in first].index("exchange")][1] == second[[e[0] for e in second].index("exchange")][1]
    # same words as one process driving the devices
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"] * world, backend=OracleBackend(), **FUSED)
    want = eng.rotate_single(synth.ciphertext(eng, 3, 0), synth.key_switch_key(eng, 6, origin="rotation key:1"))
    for r in range(world):
        assert (outs[r][0] == want.data[0][r].numpy()).all() and (outs[r][1] == want.data[1][r].numpy()).all()


# ---- BASELINE configs[3] in shape: the gold chain (34 scale primes + base + 4 special = 39 limbs, dnum 10) over 8 ranks ----
GOLD_SHAPE = dict(logN=13, num_scales=34, num_special_primes=4, is_secured=False)


def _chain_ops(eng, synth):
    """Level 0 (8 ranks alive, rows 11/8/../8 with the special limbs), the rescale 9 -> 10 after which rank 7 holds
    nothing, level 20 (4 alive) and level 32 -> 33 (rank 0 alone): rns_partition(35, 4, 8)."""
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    out = {}
    a0, b0 = synth.ciphertext(eng, 40, 0), synth.ciphertext(eng, 41, 0)
    out["mult0"] = eng.cc_mult(a0, b0, evk)
    out["rot0"] = eng.rotate_single(a0, rotk)
    a9, b9 = synth.ciphertext(eng, 42, 9), synth.ciphertext(eng, 43, 9)
    out["mult9"] = eng.cc_mult(a9, b9, evk)                       # level 9 -> 10: rank 7 drops out
    out["rot10"] = eng.rotate_single(out["mult9"], rotk)
    out["rot10b"] = eng.rotate_single(out["rot10"], rotk)         # the digit buffer and its message list again
    a20 = synth.ciphertext(eng, 44, 20)
    out["rot20"] = eng.rotate_single(a20, rotk)
    out["mult20"] = eng.cc_mult(a20, a20, evk)
    a32 = synth.ciphertext(eng, 45, 32)
    out["mult32"] = eng.cc_mult(a32, a32, evk)
    return out


def _chain_worker(rank, world, port, outdir, exchange="p2p"):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), comm=DistComm(local_device="cpu", exchange=exchange), **GOLD_SHAPE)
    for name, ct in _chain_ops(eng, synth).items():
        for comp, shards in enumerate(ct.data):
            arr = shards[0].numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name}.{comp}.{rank}.npy"), arr)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["p2p", "allgather"])
def test_eight_ranks_gold_chain_shape_through_levels_where_ranks_drop_out(exchange):
    """.. in both forms of the digit exchange: the all-gather is a collective of the whole group, so the ranks that have run
    out of rows at a level still take part in it (padding in, nothing kept) — word for word the point-to-point results."""
    warnings.filterwarnings("ignore")
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"] * 8, backend=OracleBackend(), **GOLD_SHAPE)
    assert [len(d) for d in eng.ntt.p.destination_arrays_with_special[0]] == [11, 8, 8, 8, 8, 8, 8, 8]
    assert (eng.len_devices[0], eng.len_devices[10], eng.len_devices[20], eng.len_devices[33]) == (8, 7, 4, 1)
    want = _chain_ops(eng, synth)
    port = 33500 + (os.getpid() % 2000) + (50 if exchange != "p2p" else 0)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_chain_worker, args=(8, port, outdir, exchange), nprocs=8, join=True)
        for name, ct in want.items():
            for comp, shards in enumerate(ct.data):
                for rank in range(8):
                    got = np.load(os.path.join(outdir, f"{name}.{comp}.{rank}.npy"))
                    exp = shards[rank].numpy() if rank < len(shards) else np.zeros((0, eng.ctx.N), dtype=np.int64)
                    assert got.shape == exp.shape and (got == exp).all(), (name, comp, rank)


# ---- balanced limb map (rns_partition(balance=True): not the reference's layout, default off) ----------------------------------
def test_balanced_limb_map_keeps_digits_and_the_order_in_which_devices_run_out_of_rows():
    from liberate_fhe_amd.ntt.rns_partition import rns_partition
    ref, bal = rns_partition(35, 4, 8), rns_partition(35, 4, 8, balance=True)
    assert [len(r) for r in ref.destination_arrays[0]] == [7, 4, 4, 4, 4, 4, 4, 4]            # the reference's gold / 8 layout
    assert [len(r) for r in bal.destination_arrays[0]] == [5, 6, 4, 4, 4, 4, 4, 4]
    assert bal.partitions == ref.partitions                                                   # same digits: same key, same words
    assert sorted(x for rows in bal.destination_arrays[0] for x in rows) == list(range(35))
    assert 34 in bal.destination_arrays[0][0] and bal.destination_arrays[0][1][-2:] == [32, 33]
    for shape in ((35, 4, 8), (35, 4, 4), (35, 4, 2), (17, 2, 2), (17, 2, 4), (6, 2, 2), (6, 2, 3), (80, 6, 8)):
        a, b = rns_partition(*shape), rns_partition(*shape, balance=True)
        L, K, D = shape
        assert max(len(r) for r in b.destination_arrays[0]) <= max(len(r) for r in a.destination_arrays[0])
        for lvl in range(L):                                   # alive devices are always 0 .. n - 1 (the engine's per-level lists)
            alive = [len(r) > K for r in b.destination_arrays_with_special[lvl]]
            assert alive == [True] * sum(alive) + [False] * (D - sum(alive)), (shape, lvl)
            assert len(b.destination_arrays[lvl]) == sum(alive)
            assert sorted(x for rows in b.destination_arrays[lvl] for x in rows) == list(range(lvl, L))
        assert [r[-K:] for r in b.destination_arrays_with_special[0]] == [list(range(L, L + K))] * D
    # max bytes per link of the point-to-point digit exchange at level 0, N = 65536: rows of the largest owner x 512 KiB
    assert max(len(r) for r in bal.destination_arrays[0]) * 65536 * 8 == 6 * 524288 < 7 * 524288


def _balanced_worker(rank, world, port, outdir):
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
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), comm=DistComm(local_device="cpu"), balanced_limb_map=True, **PARAMS)
    assert [len(r) for r in eng.ntt.p.destination_arrays[0]] == [3, 3]            # the reference's layout: 4 / 2
    for name, ct in _ops(eng, synth).items():
        dest = eng.ntt.p.destination_arrays[ct.level]
        for comp, shards in enumerate(ct.data):
            arr = shards[0].numpy() if shards else np.zeros((0, eng.ctx.N), dtype=np.int64)
            np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.npy"), arr)
            if rank < len(dest):
                np.save(os.path.join(outdir, f"{name.replace('/', '_')}.{comp}.{rank}.primes.npy"), np.array(dest[rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_the_balanced_limb_map_hold_the_one_device_words_limb_by_limb():
    """small over 2 ranks with balanced_limb_map=True (3 / 3 rows instead of 4 / 2): rescale fan-out, digit exchange, deeper
    levels — every limb's canonical words equal the single-device engine's (the reference digests are per device layout, so the
    comparison is by prime index)."""
    warnings.filterwarnings("ignore")
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    one = ckks_engine(devices=["cpu"], backend=OracleBackend(), **PARAMS)
    want = {name.replace("/", "_"): ct for name, ct in _ops(one, synth).items()}
    port = 29500 + (os.getpid() % 2000) + 171
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_balanced_worker, args=(2, port, outdir), nprocs=2, join=True)
        for name, ct in want.items():
            first = one.ntt.p.destination_arrays[ct.level][0][0]
            for comp in range(2):
                seen = 0
                for rank in range(2):
                    f = os.path.join(outdir, f"{name}.{comp}.{rank}.primes.npy")
                    if not os.path.exists(f):
                        continue
                    primes, rows = np.load(f), np.load(os.path.join(outdir, f"{name}.{comp}.{rank}.npy"))
                    assert rows.shape[0] == len(primes)
                    for r, prime in enumerate(primes):
                        assert (rows[r] == ct.data[comp][0][prime - first].numpy()).all(), (name, comp, rank, int(prime))
                        seen += 1
                assert seen == ct.data[comp][0].shape[0], (name, comp)


def _balanced_decrypt_worker(rank, world, port, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from liberate_fhe_amd.fhe import ckks_engine
    from liberate_fhe_amd.fhe.comm import DistComm
    from tests.oracle_backend import OracleBackend
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), comm=DistComm(local_device="cpu"), balanced_limb_map=True, **PARAMS)
    assert eng._scaler_dev[:4] == [0, 0, 1, 1]
    sk = eng.create_secret_key()
    pk = eng.create_public_key(sk)
    np.random.seed(5)
    m = eng.example(-1, 1)
    out = {}
    for level in (0, 1, 2, 3):
        ct = eng.encorypt(m, pk, level=level)
        got = eng.decrode(ct, sk)                      # level >= 2: rank 1 computes the scaler row and sends it to rank 0
        assert (got is None) == (rank != 0)
        if rank == 0:
            out[level] = float(np.abs(got - m).max())
    if rank == 0:
        with open(os.path.join(outdir, "err.json"), "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_decrypt_in_the_balanced_limb_map_with_the_scaler_row_from_rank_one():
    """From the level where device 0 holds only the base prime (small / 2 ranks: level 2) the final scaling's scale-prime row is
    the first row of rank 1: both ranks call decrode, rank 1 sends that one row, rank 0 returns the plaintext."""
    port = 29500 + (os.getpid() % 2000) + 173
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_balanced_decrypt_worker, args=(2, port, outdir), nprocs=2, join=True)
        err = json.load(open(os.path.join(outdir, "err.json")))
    assert set(err) == {"0", "1", "2", "3"} and all(v < 1e-5 for v in err.values()), err
