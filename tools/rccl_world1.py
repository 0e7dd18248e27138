"""RCCL at world size 1 on ONE GPU: everything about the limb-sharded engine's communication that a one-GPU lease can execute
against the real library — library load, communicator creation (HSA_ENABLE_IPC_MODE_LEGACY=0), the collectives and
point-to-point batches fhe/comm.py issues, the stream-ordering semantics the overlap relies on, and the engine's SHARDED
code path (native halves, both exchange forms, eager launches and HIP-graph replay) on top of an RCCL communicator of one
rank (DistComm(solo_sharded=True)) — against the unsharded engine, word for word.  What it cannot show is xGMI itself.

Run as a FRESH process (never re-exec one that touched the GPU): `python tools/rccl_world1.py [--preset gold]`.  Prints one
JSON line; exit code 0 = every check ok.  tests/test_distributed_gpu.py and bench.py (`comm.rccl_world1`) run it as a child.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="gold")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-engine", action="store_true")
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    import __graft_entry__ as g
    g.build()                                         # a no-op when libckks_hip.so is current; never touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import datetime
    import torch
    import torch.distributed as dist

    out = {"backend": None, "world_size": 1, "checks": {}, "engine": {}, "ok": False,
           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    dev = "cuda:0"
    torch.cuda.set_device(0)

    def verdict(name, fn, into=None):
        t0 = time.perf_counter()
        rec = {}
        try:
            res = fn()
            rec["ok"] = bool(res if not isinstance(res, dict) else res.pop("ok"))
            if isinstance(res, dict):
                rec.update(res)
        except Exception as e:
            rec["ok"] = False
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        torch.cuda.synchronize()
        rec["ms"] = round(1e3 * (time.perf_counter() - t0), 2)
        (out["checks"] if into is None else into)[name] = rec
        return rec["ok"]

    def bring_up():
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=300))
        probe = torch.ones(4, dtype=torch.int64, device=dev)
        dist.all_reduce(probe)                       # the first collective creates the communicator
        torch.cuda.synchronize()
        out["backend"] = str(dist.get_backend())
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out["rccl_version"] = None
        return bool((probe == 1).all().item())

    if not verdict("communicator", bring_up):
        print(json.dumps(out), flush=True)
        return 1

    W = 4096

    def all_gather_into_tensor():
        src = torch.arange(7 * W, dtype=torch.int64, device=dev).view(7, W)
        dst = torch.full((7, W), -1, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        return torch.equal(src, dst)

    def broadcast():
        t = torch.arange(10, dtype=torch.int64, device=dev) * 3
        dist.broadcast(t, src=0)
        torch.cuda.synchronize()
        return torch.equal(t, torch.arange(10, dtype=torch.int64, device=dev) * 3)

    def self_addressed_batch():
        a = torch.arange(5 * W, dtype=torch.int64, device=dev).view(5, W)
        b = torch.full((5, W), -1, dtype=torch.int64, device=dev)
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        return torch.equal(a, b)

    def comm_patterns():
        """The calls of fhe/comm.py themselves at one rank: fan-out and point-to-point exchange = EMPTY batches (no peer),
        the all-gather form = one real collective whose slab must come back as the rows it was packed from."""
        from liberate_fhe_amd.fhe.comm import DistComm
        c = DistComm(local_device=dev)
        buf = torch.arange(11 * W, dtype=torch.int64, device=dev).view(11, W)
        keep = buf.clone()
        pieces = [(0, 0, 7), (0, 7, 4)]
        c.fanout_into(buf[:4], 0, [0])
        h = c.exchange_rows(buf, pieces, [0])
        empty = len(h.works) == 0
        h.wait()
        g = DistComm(local_device=dev, exchange="allgather")
        hg = g.exchange_rows(buf, pieces, [0])
        hg.wait()
        torch.cuda.synchronize()
        send, recv = next(iter(g._slabs.values()))
        return {"ok": empty and torch.equal(buf, keep) and torch.equal(recv[0], keep), "p2p_batch_messages": len(h.works),
                "broadcast_int": c.broadcast_int(41) == 41}

    def stream_ordering():
        """(1) a collective issued behind a slow producer on the caller's stream must read what the producer wrote;
        (2) after work.wait() a consumer kernel on the caller's stream must read what the collective wrote — with async_op
        and without, on the default stream and on a side stream: the dependencies the engine's overlap rests on."""
        res = {}
        for label, stream in (("default_stream", None), ("side_stream", torch.cuda.Stream(device=dev))):
            for async_op in (True, False):
                ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
                with ctx:
                    src = torch.zeros((64, W), dtype=torch.int64, device=dev)
                    dst = torch.full((64, W), -1, dtype=torch.int64, device=dev)
                    torch.cuda.synchronize()
                    torch.cuda._sleep(int(2e8))                       # ~0.1 s of device time in front of the producer
                    src.fill_(42)
                    work = dist.all_gather_into_tensor(dst, src, async_op=async_op)
                    if async_op:
                        work.wait()
                    after = dst * 2                                   # consumer on the caller's stream
                    torch.cuda._sleep(int(1e7))
                torch.cuda.synchronize()
                res[f"{label}{'_async' if async_op else ''}"] = bool((after == 84).all().item())
        # a point-to-point batch to oneself the same way
        a = torch.zeros((16, W), dtype=torch.int64, device=dev)
        b = torch.full((16, W), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2e8))
        a.fill_(7)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
            w.wait()
        after = b + 1
        torch.cuda.synchronize()
        res["self_p2p_batch"] = bool((after == 8).all().item())
        res["ok"] = all(res.values())
        return res

    verdict("all_gather_into_tensor", all_gather_into_tensor)
    verdict("broadcast", broadcast)
    verdict("batch_isend_irecv_self_addressed", self_addressed_batch)
    verdict("comm_patterns_one_rank", comm_patterns)
    verdict("stream_ordering", stream_ordering)

    # ---- the engine's sharded code path on the RCCL communicator ------------------------------------------------------
    if not args.no_engine:
        from liberate_fhe_amd.fhe import ckks_engine, presets
        from liberate_fhe_amd.fhe.comm import DistComm
        from liberate_fhe_amd.utils import synth
        params = {k: v for k, v in presets.params[args.preset].items() if k != "devices"}
        calls = {"all_gather_into_tensor": 0, "broadcast": 0, "batch_isend_irecv": 0}
        real = {k: getattr(dist, k) for k in calls}

        def counted(name):
            def f(*a, **kw):
                calls[name] += 1
                return real[name](*a, **kw)
            return f
        for k in calls:
            setattr(dist, k, counted(k))

        def ops(eng, evk, rotk, deep):
            a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
            res = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk)]
            if deep:
                a9, b9 = synth.ciphertext(eng, 7, deep), synth.ciphertext(eng, 8, deep)
                res.append(eng.cc_mult(a9, b9, evk))
                res.append(eng.rotate_single(res[-1], rotk))
            return res

        ref_eng = ckks_engine(devices=[dev], **params)
        deep = min(9, ref_eng.num_levels - 3)
        evk0 = synth.key_switch_key(ref_eng, 5)
        rotk0 = synth.key_switch_key(ref_eng, 6, origin="rotation key:1")
        want = ops(ref_eng, evk0, rotk0, deep)
        a0, b0 = synth.ciphertext(ref_eng, 3, 0), synth.ciphertext(ref_eng, 4, 0)

        def timed(fn, n=30):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3

        out["engine"]["preset"] = args.preset
        out["engine"]["unsharded_cc_mult_us"] = round(timed(lambda: ref_eng.cc_mult(a0, b0, evk0)), 1)

        for mode in ("p2p", "allgather"):
            def run(mode=mode):
                before = dict(calls)
                eng = ckks_engine(devices=[dev], comm=DistComm(local_device=dev, exchange=mode, solo_sharded=True), **params)
                assert eng._multi and eng.local_ids == [0]
                evk = synth.key_switch_key(eng, 5)
                rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
                rec = {}
                bad = 0
                for label, graphs, reps in (("eager", False, 1), ("graph", True, args.reps)):
                    eng.graph_sharded = graphs
                    for _ in range(reps):            # graphs: the first call of each (kind, level) runs eagerly and captures,
                        got = ops(eng, evk, rotk, deep)   # the later ones replay around the live exchange
                    n_bad = sum(0 if torch.equal(g.data[c][0], w.data[c][0]) else 1 for g, w in zip(got, want) for c in range(2))
                    rec[f"{label}_rows_differing"] = n_bad
                    bad += n_bad
                graphs_held = sum(1 for k in eng._tables if isinstance(k, tuple) and k and k[0] == "sgraph")
                rec["graph_segment_sets"] = graphs_held
                a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
                rec["cc_mult_us_graph_replay"] = round(timed(lambda: eng.cc_mult(a, b, evk)), 1)
                eng.graph_sharded = False
                rec["cc_mult_us_eager"] = round(timed(lambda: eng.cc_mult(a, b, evk)), 1)
                rec["backend_calls"] = {k: calls[k] - before[k] for k in calls}
                took = rec["backend_calls"]
                issued = took["all_gather_into_tensor"] > 0 if mode == "allgather" else True
                rec["ok"] = bad == 0 and graphs_held >= 2 and issued and took["broadcast"] >= 1
                return rec
            verdict(f"sharded_path_{mode}", run, into=out["engine"])

    out["ok"] = all(v.get("ok") for v in out["checks"].values()) and all(v.get("ok") for v in out["engine"].values() if isinstance(v, dict))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
