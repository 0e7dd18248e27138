"""rns_partition(balance=True) against the reference's limb map on the slowest rank of a gold engine sharded over 8 GPUs, measured the
way profiles/r05_host_overhead.txt was: ONE GPU, a stand-in communicator whose exchanges return at once (tools/host_overhead_null.py:
launches and Python are the real ones, results are not), device-paced time per op of rank 0 and rank 1 in both maps, plus the rows each
rank holds and the bytes its digit batch puts on a link (bench.link_bytes: rows of the owner x N x 8).
    python tools/balanced_map_probe.py > profiles/r06_balanced_map.txt"""
import os, sys, time, warnings
import torch
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
from tools.host_overhead_null import NullComm
import bench

params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
print("# gold, limb-sharded over 8 ranks, exchanges stubbed (one GPU), eager launches; device-paced us/op = median of 5 bursts of 8")
for balanced in (False, True):
    for rank in (0, 1, 2):
        eng = ckks_engine(devices=["cuda:0"], comm=NullComm(8, rank), balanced_limb_map=balanced, **params)
        eng.graph_sharded = False
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        lb = bench.link_bytes(eng, 0)
        rows = [len(r) for r in eng.ntt.p.destination_arrays[0]]
        line = [f"map {'balanced ' if balanced else 'reference'} rank {rank}: rows per rank {rows}, this rank {rows[rank]} (+4 special)",
                f"max bytes per link {lb['max_bytes_per_link'] / 1e6:.2f} MB", f"messages per batch on this rank {lb['messages_per_batch_per_rank'][rank]}"]
        for name, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate", lambda: eng.rotate_single(a, rotk))):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 8)
            ts.sort()
            line.append(f"{name} {1e6 * ts[2]:.1f} us")
        print(" | ".join(line), flush=True)
        del eng, a, b, evk, rotk
        torch.cuda.empty_cache()
