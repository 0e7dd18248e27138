"""Host time per engine op (time to ENQUEUE an op, no synchronisation inside the loop) against its device time:
the margin the one-process-per-GPU sharded path has before the host, not the GPU, paces it.  Short bursts from an idle
queue (the launch queue must not fill, or the enqueue time tracks the device)."""
import sys, time, warnings
import torch
warnings.filterwarnings("ignore")
sys.path.insert(0, ".")
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth

for preset in ("silver", "gold"):
    params = {k: v for k, v in presets.params[preset].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0"], **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    pinv = pow(5, -1, 2 * eng.ctx.N)
    plan, _, fp, ro = eng._op_plan(0, 0)
    kp = eng._key_pack(rotk)[0]
    out = torch.empty((2, plan.ell, eng.ctx.N), dtype=torch.int64, device="cuda:0")
    raw = lambda: eng.backend.switch_key_native(plan, a.data[0][0], a.data[1][0], pinv, True, kp, fp, ro, out)
    cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(16)]
    pairs = [(cts[i], cts[(i + 1) % 16]) for i in range(16)]
    for name, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate", lambda: eng.rotate_single(a, rotk)),
                     ("lf_switch_key alone", raw),
                     ("rotate_single_batch of 16 (4 groups of 4, two lanes; per call)", lambda: eng.rotate_single_batch(cts, rotk)),
                     ("cc_mult_batch of 16 (4 groups of 4, two lanes; per call)", lambda: eng.cc_mult_batch(pairs, evk))):
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        host, total, n = [], [], 8
        for rep in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) / n)
            total.append((t2 - t0) / n)
        host.sort(); total.sort()
        print(f"{preset} {name}: host enqueue {1e6 * host[len(host) // 2]:.1f} us/op (median of 10 bursts of {n}), "
              f"device-paced {1e6 * total[len(total) // 2]:.1f} us/op")
    del eng
    torch.cuda.empty_cache()


# ---- a rank of a limb-sharded op: host time to enqueue it, exchanges stubbed out ------------------------------------
# (no second GPU here: a stand-in communicator whose exchanges return at once — the words a rank would receive are
# whatever its buffers hold, so the RESULTS are meaningless; the launches and the Python around them are the real ones)
from tools.host_overhead_null import NullComm  # noqa: E402


for world in (2, 8):
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}
    eng = ckks_engine(devices=["cuda:0"], comm=NullComm(world), **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for graphs in (False, True):
      eng.graph_sharded = graphs
      for name, fn in (("cc_mult", lambda: eng.cc_mult(a, b, evk)), ("rotate", lambda: eng.rotate_single(a, rotk))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        host, n = [], 8
        for rep in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            host.append((time.perf_counter() - t0) / n)
            torch.cuda.synchronize()
        host.sort()
        rows = len(eng.ntt.p.destination_arrays_with_special[0][0])
        dev_t = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dev_t.append((time.perf_counter() - t0) / n)
        dev_t.sort()
        print(f"gold {name}, rank 0 of {world} (limb-sharded, {rows} of 39 rows, exchanges stubbed), fixed-address launches "
              f"{'replayed from 3 HIP graphs' if graphs else 'eager'}: host enqueue {1e6 * host[len(host) // 2]:.1f} us/op, device-paced {1e6 * dev_t[2]:.1f} us/op")
    del eng
    torch.cuda.empty_cache()
