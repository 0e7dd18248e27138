#!/usr/bin/env python3
"""Headline benchmark: NTTs/sec/GPU at logN=16, L=30 limbs (+ cc_mult_evk ops/sec), % of HBM roofline.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

One "step" = one forward negacyclic NTT (ntt_cuda.ntt semantics, bit-exact lazy outputs) of a batch of
B polynomials x 30 RNS limbs at logN = 16 — the gold preset's with-special row set at level 9 (25 scale
primes + base + 4 special primes).  B = 128 by default (3.75 GiB-class working set: B x 30 MiB, far beyond the
256 MiB Infinity Cache; at B = 16 the column pass still finds part of its input there and the launch tail of the
tiled pass is 10 block rounds instead of 80).  Inputs are synthetic (splitmix64 -> mod 2q), resident in HBM before the timed region.
Polynomials are independent, so ranks shard them with no data-path collective: weak scaling.

The JSON line also carries
  roofline     : the dominant kernel (ntt_pass16_fwd_seq: the tiled pass of all 30 limbs, the second of the two
                 launches of a transform, 12 of its 16 stages) against the 8 TB/s HBM peak; algorithmic bytes
                 per launch = 8*N*limbs (a transform is 16*N bytes per limb, SURVEY.md §8d, spread over its two
                 launches); its launch duration is measured live with HIP events on the launch stream, the
                 kernel launched alone (lf_ntt_pass, the measurement entry) with the grid it has inside the full step;
                 `shader_clock_mhz` = the clock the CUs run at beside each kernel (lf_clock_probe on a second stream) and
                 `package_power` = the amdgpu hwmon reading while the dominant kernel runs: both passes sit at the 1 400 W
                 package cap and the tiled pass runs at ~1.9 of 2.4 GHz — `frac` is what that clock allows;
  cpu_baseline : the C oracle (strict reference-kernel semantics) on this box's host cores, same workload,
                 bounded sample;
  roofline_engine_ops : cc_mult_evk / rotate_single per preset against the same HBM peak with SURVEY.md §8(d)'s
                 algorithmic bytes (N = 1 only);
  extra        : cc_mult(+relinearize) ops/s for silver and gold on this rank, rotate ops/s, limb-NTT/s.  At N > 1:
                 gold cc_mult as replicas and LIMB-SHARDED over the ranks (RCCL; BASELINE configs[3]) and the
                 64-ciphertext gold rotate batch (configs[4]), all timed by default under a watchdog.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0   # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s spec figure
L_LIMBS = 30
LOGN = 16


def stamped_profile(pattern):
    """The newest profiles/<pattern> (e.g. "traffic_r*.json") whose `library_digest` equals the digest of the sources the
    running libckks_hip.so is built from (__graft_entry__.library_digest); None otherwise.  PMC-derived figures describe
    the kernels they were collected on: a changed kernel must not inherit them."""
    import glob
    import __graft_entry__ as g
    want = g.library_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("library_digest") == want:
            j["_source"] = os.path.relpath(path, ROOT)
            return j
    return None


def event_time_ms(fn, iters):
    """Average duration of fn() over `iters` calls, HIP events on torch's current stream (the stream
    every C-ABI launch in this process uses)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(ctx, rows_idx, batch, budget_s=12.0):
    """The oracle's NTT on the host cores over the same [batch*30, N] stack (OpenMP over rows)."""
    from oracle import oracle as orc
    h = lambda v: np.ascontiguousarray(np.tile(np.asarray([v[i] for i in rows_idx], dtype=np.int64), batch))
    ql, qh, kl, kh = h(ctx.q_lower_bits), h(ctx.q_higher_bits), h(ctx.k_lower_bits), h(ctx.k_higher_bits)
    q2, Rs = h(ctx.q_double), h(ctx.R_square)
    psi = np.ascontiguousarray(np.tile(ctx.psi_br[rows_idx], (batch, 1)))
    n = len(rows_idx) * batch
    orc.mont_enter(psi, Rs, n, ql, qh, kl, kh)
    rng = np.random.default_rng(5)
    x = rng.integers(0, 1 << 40, size=(n, ctx.N), dtype=np.int64)
    orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)   # warm (page-in, thread pool)
    t0, reps = time.time(), 0
    while True:
        orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)
        reps += 1
        if time.time() - t0 > budget_s or reps >= 50:
            break
    dt = (time.time() - t0) / reps
    threads = min(os.cpu_count() or 1, n)
    out = {"value": batch / dt, "unit": "poly-NTT(L=30,logN=16)/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x forward NTT of {batch} polys x {len(rows_idx)} limbs, N=65536, C oracle + OpenMP over limb rows"}
    # the same kernel on ONE core: one polynomial's 30 limbs, OpenMP pinned to a single thread
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        before = gomp.omp_get_max_threads()
        gomp.omp_set_num_threads(1)
        n1 = len(rows_idx)
        x1 = np.ascontiguousarray(x[:n1])
        orc.ntt(x1, psi[:n1], n1, ctx.logN, q2[:n1], ql[:n1], qh[:n1], kl[:n1], kh[:n1])
        t0, reps1 = time.time(), 0
        while True:
            orc.ntt(x1, psi[:n1], n1, ctx.logN, q2[:n1], ql[:n1], qh[:n1], kl[:n1], kh[:n1])
            reps1 += 1
            if time.time() - t0 > 4.0 or reps1 >= 20:
                break
        gomp.omp_set_num_threads(before)
        out["single_thread"] = {"value": reps1 / (time.time() - t0), "unit": "poly-NTT(L=30,logN=16)/s", "cores": 1,
                                "sample": f"{reps1} x forward NTT of 1 poly x {n1} limbs, one OpenMP thread"}
    except Exception as e:   # a baseline, never a reason to lose the line
        out["single_thread"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def cpu_ntt_baseline_preset(name, budget_s=4.0):
    """the same oracle transform at the other presets' shapes — every limb of the preset's chain
    (special primes included), ONE polynomial, all cores (OpenMP over limb rows)."""
    from oracle import oracle as orc
    from liberate_fhe_amd.fhe import presets
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    params = {k: v for k, v in presets.params[name].items() if k != "devices"}
    ctx = ckks_context(**params)
    n = len(ctx.q)
    h = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.int64))
    ql, qh, kl, kh = h(ctx.q_lower_bits), h(ctx.q_higher_bits), h(ctx.k_lower_bits), h(ctx.k_higher_bits)
    q2, Rs = h(ctx.q_double), h(ctx.R_square)
    psi = np.ascontiguousarray(ctx.psi_br.copy())
    orc.mont_enter(psi, Rs, n, ql, qh, kl, kh)
    x = np.random.default_rng(6).integers(0, 1 << 40, size=(n, ctx.N), dtype=np.int64)
    orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)
    t0, reps = time.time(), 0
    while True:
        orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)
        reps += 1
        if time.time() - t0 > budget_s or reps >= 200:
            break
    dt = (time.time() - t0) / reps
    return {"value": 1.0 / dt, "unit": f"poly-NTT(L={n},logN={ctx.logN})/s", "limb_ntt_per_s": n / dt, "cores": min(os.cpu_count() or 1, n),
            "kind": "port", "sample": f"{reps} x forward NTT of 1 polynomial x {n} limbs, N={ctx.N}, C oracle + OpenMP over limb rows"}


def algorithmic_rows(eng, op):
    """rows of N*8 bytes an op has to move at level 0 -> 1 (cc_mult) or level 0 (rotate).
    cc_mult : read 4(l+1) input rows + 2 dnum (l+K) key rows, write 2 l rows, l = rows after the rescale;
    rotate  : read 2 l' + 2 dnum (l'+K) key rows, write 2 l' rows, l' = rows at level 0."""
    K = eng.ntt.num_special_primes
    if op == "cc_mult":
        ell = len(eng.ntt.p.destination_arrays[1][0])
        dnum = len(eng._ks_tables(1)["order"])
        return 4 * (ell + 1) + 2 * dnum * (ell + K) + 2 * ell
    ell = len(eng.ntt.p.destination_arrays[0][0])
    dnum = len(eng._ks_tables(0)["order"])
    return 2 * ell + 2 * dnum * (ell + K) + 2 * ell


def op_roofline(eng, op, ops_per_s, profile):
    rows = algorithmic_rows(eng, op)
    nbytes = rows * eng.ctx.N * 8
    achieved = nbytes * ops_per_s / 1e9
    out = {"bound": "hbm", "algorithmic_rows": rows, "algorithmic_bytes": nbytes, "ops_per_s": ops_per_s,
           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "frac_of_achievable_6300": achieved / HBM_ACHIEVABLE_GBS}
    if profile:
        out["from_profile"] = profile    # dominant kernel + per-kernel time, rocprofv3 summary under profiles/
    return out


def engine_rates(dev, quick):
    """cc_mult(+relinearize) and rotate_single ops/s on this rank for silver and gold (1 GPU each), each with its
    HBM roofline block (algorithmic bytes of SURVEY.md §8(d))."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    out = {}
    roof = {}
    prof = stamped_profile("r*_engine_ops_summary.json") or {}   # per-kernel times: only if taken on this build
    for name in ("silver", "gold"):
        eng = ckks_engine(**{**presets.params[name], "devices": [dev]})
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        # steady state: the first ~10 ms after the set-up phase (host-to-device copies, table builds) run at a lower
        # clock — gold cc_mult measures 588 us/op over ops 4-13 and 538 us/op from op 50 on (tools/warm_probe.py) —
        # so every rate below is taken after 40 untimed ops, over 100 (silver) / 60 (gold) timed ones
        for _ in range(40):
            eng.cc_mult(a, b, evk)
            eng.rotate_single(a, rotk)
        torch.cuda.synchronize()
        n = 5 if quick else (100 if name == "silver" else 60)
        ms = event_time_ms(lambda: eng.cc_mult(a, b, evk), n)
        out[f"cc_mult_evk_{name}_ops_per_s"] = 1e3 / ms
        roof[f"cc_mult_evk_{name}"] = op_roofline(eng, "cc_mult", 1e3 / ms, prof.get(f"{name}_cc_mult"))
        ms = event_time_ms(lambda: eng.rotate_single(a, rotk), n)
        out[f"rotate_single_{name}_ops_per_s"] = 1e3 / ms
        roof[f"rotate_single_{name}"] = op_roofline(eng, "rotate", 1e3 / ms, prof.get(f"{name}_rotate"))
        # configs[4]: a batch of ciphertexts rotated by the same step (one key): groups of 4 per key-switch launch set
        nb = 16
        cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(nb)]
        for _ in range(3):
            eng.rotate_single_batch(cts, rotk)
        torch.cuda.synchronize()
        ms = event_time_ms(lambda: eng.rotate_single_batch(cts, rotk), max(2, n // 8))
        out[f"rotate_single_{name}_batch{nb}_rotations_per_s"] = nb * 1e3 / ms
        pairs = [(cts[i], cts[(i + 1) % nb]) for i in range(nb)]
        for _ in range(3):
            eng.cc_mult_batch(pairs, evk)
        torch.cuda.synchronize()
        ms = event_time_ms(lambda: eng.cc_mult_batch(pairs, evk), max(2, n // 8))
        out[f"cc_mult_evk_{name}_batch{nb}_ops_per_s"] = nb * 1e3 / ms
        roof[f"cc_mult_evk_{name}_batch{nb}"] = op_roofline(eng, "cc_mult", nb * 1e3 / ms, None)
        if name == "gold":
            # BASELINE configs[4] at its stated size: 64 level-0 ciphertexts (seeds 100..163) under one rotation key
            cts64 = cts + [synth.ciphertext(eng, 100 + i, 0) for i in range(nb, 64)]
            eng.rotate_single_batch(cts64, rotk)
            torch.cuda.synchronize()
            ms = event_time_ms(lambda: eng.rotate_single_batch(cts64, rotk), 3)
            out["rotate_single_gold_batch64_rotations_per_s"] = 64 * 1e3 / ms
            roof["rotate_single_gold_batch64"] = op_roofline(eng, "rotate", 64 * 1e3 / ms, None)
            del cts64
        del cts, pairs
        del eng, a, b, evk, rotk
        torch.cuda.empty_cache()
    return out, roof


def cpu_engine_baseline(preset="silver", budget_s=25.0, max_reps=10):
    """cc_mult(+relinearize) of a preset on the host cores: this package's orchestration over the CHECKER backend
    (tests/oracle_backend.py: the reference's composition of ntt_cuda calls, each call the C oracle with OpenMP
    over limb rows, torch elementwise ops in between — what the reference engine would do if its extension were
    a CPU library).  Bounded sample: as many ops as fit the budget, at least one."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    from tests.oracle_backend import OracleBackend
    params = {k: v for k, v in presets.params[preset].items() if k != "devices"}
    eng = ckks_engine(devices=["cpu"], backend=OracleBackend(), **params)
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    eng.cc_mult(a, b, evk)       # warm: tables, thread pool
    t0, reps = time.time(), 0
    while True:
        eng.cc_mult(a, b, evk)
        reps += 1
        if time.time() - t0 > budget_s or reps >= max_reps:
            break
    dt = (time.time() - t0) / reps
    return {"value": 1.0 / dt, "unit": f"cc_mult_evk({preset})/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"{reps} x cc_mult+relinearize, {preset} (logN {eng.ctx.logN}, level 0 -> 1), checker backend (C oracle + OpenMP, "
                      f"reference-shaped orchestration), after one warm-up op"}


def _max_over_ranks(ms, dev):
    import torch.distributed as dist
    t = torch.tensor([ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _natural(eng, ct):
    """{prime index: [comp 0 row, comp 1 row]} of the rows THIS process holds of a ciphertext."""
    dest = eng.ntt.p.destination_arrays[ct.level]
    out = {}
    for i, d in enumerate(dd for dd in eng.local_ids if dd < len(dest)):
        for r, prime in enumerate(dest[d]):
            out[prime] = [ct.data[0][i][r], ct.data[1][i][r]]
    return out


def multi_gpu_rates(dev, world, rank, out, sharded=True):
    """N > 1.  Fills `out` (a dict that main() prints even if a leg below never returns, see the watchdog):
      replicas      gold cc_mult(+relinearize) on N independent engines, zero communication: N x single rate;
      config 5      64 gold ciphertexts rotated under one key, 64 / N per rank (batch replicas, full key per GPU);
      config 4      gold cc_mult LIMB-SHARDED over the N ranks by rns_partition: the dropped limb's rows fan out point
                    to point from their owner, the key-switch digits travel as ONE batch of point-to-point messages
                    (RCCL send / recv over xGMI) while each rank extends + transforms the digits it owns (fhe/comm.py,
                    lf_ks_fwd / lf_ks_tail); plus limb-sharded rotate_single.
    The sharded legs are PARITY-GATED: before anything is timed every rank compares the rows it holds of the sharded
    results — level 0 -> 1, and 9 -> 10 + two key switches at level 10, where the last rank has run out of rows —
    word for word with the unsharded engine of the replica leg on the same rank; a mismatch anywhere withholds the
    rates.  Every leg is fenced by barriers on the default group and catches its own exceptions; the sharded legs run
    on their OWN process group with a short timeout, so a rank that fails inside them cannot hang the line."""
    import datetime
    import torch.distributed as dist
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    params = {k: v for k, v in presets.params["gold"].items() if k != "devices"}

    def parity_ops(eng, evk, rotk):
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        a9, b9 = synth.ciphertext(eng, 7, 9), synth.ciphertext(eng, 8, 9)
        res = [eng.cc_mult(a, b, evk), eng.rotate_single(a, rotk), eng.cc_mult(a9, b9, evk)]
        res.append(eng.rotate_single(res[-1], rotk))
        res.append(eng.rotate_single(res[-1], rotk))      # same digit buffer, same cached message list
        return res

    want = None
    try:
        eng = ckks_engine(devices=[dev], **params)
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        if sharded:
            want = [_natural(eng, ct) for ct in parity_ops(eng, evk, rotk)]
        for _ in range(40):
            eng.cc_mult(a, b, evk)
        torch.cuda.synchronize()
        dist.barrier()
        ms = _max_over_ranks(event_time_ms(lambda: eng.cc_mult(a, b, evk), 60), dev)
        out["cc_mult_evk_gold_replicas_ops_per_s"] = world * 1e3 / ms
        nb = 16
        cts = [synth.ciphertext(eng, 100 + i, 0) for i in range(nb)]
        pairs = [(cts[i], cts[(i + 1) % nb]) for i in range(nb)]
        eng.cc_mult_batch(pairs, evk)
        torch.cuda.synchronize()
        dist.barrier()
        ms = _max_over_ranks(event_time_ms(lambda: eng.cc_mult_batch(pairs, evk), 3), dev)
        out[f"cc_mult_evk_gold_replicas_batch{nb}_ops_per_s"] = world * nb * 1e3 / ms
        del pairs
        # config 5, batch-replica mode: ciphertexts 100..163, rank r takes every world-th one
        mine = [synth.ciphertext(eng, 100 + i, 0) for i in range(rank, 64, world)]
        eng.rotate_single_batch(mine, rotk)
        torch.cuda.synchronize()
        dist.barrier()
        ms = _max_over_ranks(event_time_ms(lambda: eng.rotate_single_batch(mine, rotk), 3), dev)
        out["rotate_single_gold_batch64_replicas_rotations_per_s"] = 64 * 1e3 / ms
        del eng, a, b, evk, rotk, cts, mine
        torch.cuda.empty_cache()
    except Exception as e:
        out["multi_gpu_replicas_error"] = f"{type(e).__name__}: {e}"[:300]
    if not sharded:
        return
    try:
        from liberate_fhe_amd.fhe.comm import DistComm
        # (its timeout must NOT undercut the watchdog below: a communicator that times out first tears the process down
        # before the line is printed)
        grp = dist.new_group(ranks=list(range(world)), timeout=datetime.timedelta(seconds=900))
        eng = ckks_engine(devices=[dev], comm=DistComm(group=grp, local_device=dev), **params)
        evk = synth.key_switch_key(eng, 5)
        rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
        # parity gate (every rank, every row it holds)
        bad = 1 if want is None else 0
        if want is not None:
            for ref, ct in zip(want, parity_ops(eng, evk, rotk)):
                for prime, rows in _natural(eng, ct).items():
                    for comp in range(2):
                        if not torch.equal(rows[comp], ref[prime][comp]):
                            bad += 1
        t = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp)
        if int(t.item()):
            out["multi_gpu_limb_sharded_error"] = (f"parity gate: {int(t.item())} row(s) of the limb-sharded results differ from the "
                                                   "unsharded engine (or no reference was computed); rates withheld")
            return
        out["limb_sharded_parity"] = ("bit-exact vs the unsharded engine on every rank: cc_mult 0->1, rotate at 0, cc_mult 9->10, "
                                      "two rotations at level 10 (at 8 ranks the last one holds no rows there)")
        a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
        for _ in range(20):
            eng.cc_mult(a, b, evk)
        torch.cuda.synchronize()
        dist.barrier(group=grp)
        ms = _max_over_ranks(event_time_ms(lambda: eng.cc_mult(a, b, evk), 40), dev)
        out["cc_mult_evk_gold_limb_sharded_ops_per_s"] = 1e3 / ms
        for _ in range(10):
            eng.rotate_single(a, rotk)
        torch.cuda.synchronize()
        dist.barrier(group=grp)
        ms = _max_over_ranks(event_time_ms(lambda: eng.rotate_single(a, rotk), 40), dev)
        out["rotate_single_gold_limb_sharded_ops_per_s"] = 1e3 / ms
        out["limb_sharded_rows_per_rank_level0"] = [len(d) for d in eng.ntt.p.destination_arrays_with_special[0]]
    except Exception as e:   # the headline line must survive a failure of this leg
        out["multi_gpu_limb_sharded_error"] = f"{type(e).__name__}: {e}"[:300]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process has made NO GPU call (importing torch makes none);
    it starts N fresh rank processes through torch.distributed.run — one per GPU, RCCL — lets rank 0's single JSON line
    through on the inherited stdout, and leaves with the launcher's exit code (non-zero if any rank failed).  Nothing
    is exec'ed over a process that has touched the GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # does not initialise the GPU
    if have < n and os.environ.get("LF_BENCH_REHEARSE") != "1":
        print(f"bench.py --gpus {n}: this box has {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="polynomials per step per GPU")
    ap.add_argument("--no-extra", action="store_true", help="skip the cc_mult / rotate / CPU legs")
    ap.add_argument("--no-sharded", action="store_true",
                    help="N > 1: skip the limb-sharded gold cc_mult / rotate legs (RCCL broadcast + all-gather), timed by default")
    ap.add_argument("--leg-timeout", type=float, default=300.0,
                    help="N > 1: seconds after which the multi-GPU engine legs are abandoned and the line is printed without them")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # development aid (never set by the driver): LF_BENCH_REHEARSE=1 runs the N > 1 code path on a ONE-GPU box —
    # every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one device).  Numbers are meaningless.
    rehearse = world > 1 and os.environ.get("LF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import datetime
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
            from tests import gloo_device_p2p      # gloo moves host memory only: device messages are staged (test transport)
            gloo_device_p2p.install()
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev), timeout=datetime.timedelta(seconds=900))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()

    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.ntt import twiddles
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context
    from liberate_fhe_amd.utils import synth

    ctx = ckks_context(logN=LOGN, num_special_primes=4)          # gold
    ntt = ntt_context(ctx, devices=[dev])
    total = len(ctx.q)
    rows_idx = list(range(total - L_LIMBS, total))
    lo = total - L_LIMBS
    N, B = ctx.N, args.batch
    x = torch.empty((B, L_LIMBS, N), dtype=torch.int64, device=dev)
    for b in range(B):
        x[b] = torch.from_numpy(synth.uniform_rows(1000 * rank + b, rows_idx, ctx.q, N, lazy=True)).to(dev)
    sl = lambda t: t[0][lo:]
    psi, q2, ql, qh, kl, kh = (sl(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
    stream = torch.cuda.current_stream().cuda_stream
    psi_dp = twiddles.dp_pointer(psi, ql, qh, kl, kh, local_rank, stream)
    q_host = np.array([ctx.q[i] for i in rows_idx], dtype=np.int64)

    ntt_flags = int(os.environ.get("LF_BENCH_NTT_FLAGS", "0"))   # experiments only: 1 = relaxed transform (not the metric)

    def step():
        check(lib.lf_ntt(x.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, ntt_flags, q2.data_ptr(), ql.data_ptr(),
                         qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / args.steps
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall / args.steps * 1e3

    value = world * B / (wall / args.steps)                         # poly-NTT(30)/s, whole job

    # Roofline of the dominant kernel, ntt_pass16_fwd_seq: the tiled pass (12 of the 16 stages) of all 30 limbs,
    # integer-class blocks first, then the fp64 class.  In the step above it follows the column pass, so it is
    # timed on its own here: lf_ntt_pass(which = 2), the library's measurement entry, launches exactly that kernel,
    # once, with the grid it has inside the full step.  (x is scratch afterwards.)
    n_roof = max(100, 2 * args.steps)      # >= 100 timed launches of the dominant kernel

    def one_pass(which):
        check(lib.lf_ntt_pass(x.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, ntt_flags, which,
                              ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt_pass")

    for _ in range(3):
        one_pass(2)
    torch.cuda.synchronize()
    k_ms = event_time_ms(lambda: one_pass(2), n_roof)
    one_pass(1)
    torch.cuda.synchronize()
    cols_ms = event_time_ms(lambda: one_pass(1), n_roof)
    # Shader clock and package power WHILE the kernels run: a one-wave probe (lf_clock_probe) on a second stream counts core
    # cycles per tick of the constant 100 MHz counter; the hwmon power file is read once in the middle of a ~1.5 s run of tiled passes.
    # (MI355X caps the package at 1 400 W; the headline kernels sit at the cap and the clock is what gives)
    def clock_under(work, launches):
        try:
            probe_out = torch.zeros(2 * 24, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize()
            for _ in range(max(1, launches // 4)):
                work()
            check(lib.lf_clock_probe(probe_out.data_ptr(), 24, 100_000, local_rank, side.cuda_stream), "lf_clock_probe")
            for _ in range(launches):
                work()
            torch.cuda.synchronize()
            o = probe_out.cpu().numpy().reshape(24, 2).astype(np.float64)
            return float(np.median(o[:, 0] / o[:, 1] * 100.0))
        except Exception:
            return None

    def read_hwmon_power():
        """(package power W, cap W) of this rank's GPU from the amdgpu hwmon files (what the SMI prints); no child process:
        under a profiler a child would inherit its preloaded library.  (None, None) when the files are not there."""
        import glob
        cards = []
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cards = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:
            pass
        cards = cards or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
        best = (None, None)
        for d in cards:
            try:
                name = "power1_average" if os.path.exists(d + "/power1_average") else "power1_input"
                cur = int(open(f"{d}/{name}").read()) / 1e6
                cap = int(open(f"{d}/power1_cap").read()) / 1e6 if os.path.exists(d + "/power1_cap") else None
                if best[0] is None or cur > best[0]:
                    best = (cur, cap)
            except Exception:
                continue
        return best

    def package_power_under(work, launches):
        try:
            torch.cuda.synchronize()
            for _ in range(launches):                      # enqueue only: the device works through them while the file is read
                work()
            time.sleep(0.7)
            pw = read_hwmon_power()
            torch.cuda.synchronize()
            return pw
        except Exception:
            torch.cuda.synchronize()
            return None, None

    clocks = power = None
    if rank == 0 and world == 1:
        clocks = {"idle": clock_under(lambda: None, 0), "tiled_pass": clock_under(lambda: one_pass(2), 40),
                  "column_pass": clock_under(lambda: one_pass(1), 60), "whole_transform": clock_under(step, 25)}
        pw, cap = package_power_under(lambda: one_pass(2), 1500)
        power = {"tiled_pass_package_W": pw, "package_cap_W": cap,
                 "at_cap": None if pw is None or cap is None else bool(pw >= 0.97 * cap),
                 "source": "amdgpu hwmon power1_average / power1_cap, read while 1 500 tiled passes run"}
    alg_bytes_per_launch = 8 * N * L_LIMBS * B                      # 16*N per limb per transform, two launches
    achieved = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
    traffic = valu = valu_busy = cols_traffic = None
    tj = stamped_profile("traffic_r*.json")   # PMC figures per launch, only if collected on THIS build (else null)
    pmc_source = None
    if tj is not None:
        pmc_source = tj["_source"]
        scale = B / float(tj.get("batch_per_gpu", 128))   # a launch's traffic is linear in the batch
        traffic = tj.get("ntt_fwd_pass_mixed_bytes_per_launch")
        traffic = None if traffic is None else traffic * scale
        cols_traffic = tj.get("ntt_fwd_cols_mixed_bytes_per_launch")
        cols_traffic = None if cols_traffic is None else cols_traffic * scale
        valu = tj.get("ntt_fwd_pass_mixed_valu_wave_instr_per_launch")
        valu = None if valu is None else valu * scale
        valu_busy = tj.get("ntt_fwd_pass_mixed_valu_busy_frac")

    result = {
        "metric": "NTTs/sec (forward negacyclic poly-NTT, logN=16, L=30 limbs, bit-exact vs reference semantics)",
        "value": value, "unit": "poly-NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"gold preset (logN=16), rows {lo}..{total - 1} of the prime chain (25 scale + base + 4 special"
                               f" primes = 30 limbs), {B} polynomials per GPU per step, forward NTT via lf_ntt (C ABI)",
                   "batch_per_gpu": B, "limbs": L_LIMBS, "logN": LOGN, "parallelism": f"replicas x{world} (independent polynomials)"},
        # What bounds the dominant kernel is VALU instruction issue, not HBM (the bytes would take half the time): `frac`
        # stays the algorithmic-bytes fraction of the 8 TB/s HBM peak the contract asks for, `issue_frac` is the measured
        # utilisation of the pipe that actually limits it (PMC; null when the counters were not taken on this build)
        "roofline": {"bound": "valu_issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "issue_frac": valu_busy, "traffic": traffic,
                     "kernel": "ntt_pass16_fwd_seq<false> (tiled pass = 12 of 16 stages, 16 words per thread, 8 tiles per block, all 30 limbs: 5 integer-class + 25 fp64-class)",
                     "launches_per_transform": 2, "avg_launch_ms": k_ms, "launches_timed": n_roof,
                     "column_pass_launch_ms": cols_ms,
                     "column_pass_algorithmic_GBps": alg_bytes_per_launch / (cols_ms * 1e-3) / 1e9,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "whole_step_algorithmic_GBps": 16 * N * L_LIMBS * B / (dev_ms * 1e-3) / 1e9,
                     "column_pass_traffic": cols_traffic,
                     # measured while the kernels run (lf_clock_probe on a second stream; idle = 2 400): the package sits at its
                     # power cap under both passes and the shader clock is what gives — `frac` is reached at THAT clock
                     "shader_clock_mhz": clocks, "package_power": power,
                     # what actually bounds the kernel: VALU issue.  PMC pass of profiles/r02_bench_pmc.txt: wave-level
                     # VALU instructions per launch (scaled to the batch) and the fraction of cycles the SIMDs' VALU is
                     # busy (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over SQ_BUSY_CU_CYCLES / 256 CUs) — a utilisation,
                     # measured, not a ratio against a synthetic peak
                     "valu": None if valu is None else {
                         "wave_instr_per_launch": valu, "wave_instr_G_per_s": valu / (k_ms * 1e-3) / 1e9,
                         "busy_frac_pmc": valu_busy, "source": pmc_source}},
    }
    extra = {"limb_ntt_per_s": value * L_LIMBS, "device_ms_per_step": dev_ms,
             "whole_step_frac_of_achievable_6300": 16 * N * L_LIMBS * B / (dev_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS}
    result["extra"] = extra
    if rehearse:
        extra["REHEARSAL"] = "all ranks on cuda:0 over gloo: not a measurement"
    if rank == 0 and world == 1 and not args.no_extra:
        # Sensitivity of the headline to the limb mix (NOT the metric): the same step on the 30 limbs of a level-5
        # ciphertext — 29 scale primes + the base prime, no special primes, i.e. 1 integer-class limb instead of 5.
        lo_ct = total - 4 - L_LIMBS
        rows_ct = list(range(lo_ct, total - 4))
        sl_ct = lambda t: t[0][lo_ct:total - 4]
        psi_c, q2_c, ql_c, qh_c, kl_c, kh_c = (sl_ct(t) for t in (ntt.psi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh))
        dp_c = twiddles.dp_pointer(psi_c, ql_c, qh_c, kl_c, kh_c, local_rank, stream)
        qh_ct = np.array([ctx.q[i] for i in rows_ct], dtype=np.int64)
        for b in range(B):
            x[b] = torch.from_numpy(synth.uniform_rows(5000 + b, rows_ct, ctx.q, N, lazy=True)).to(dev)

        def step_ct():
            check(lib.lf_ntt(x.data_ptr(), B, L_LIMBS, LOGN, psi_c.data_ptr(), dp_c, qh_ct.ctypes.data, 0, 0, q2_c.data_ptr(),
                             ql_c.data_ptr(), qh_c.data_ptr(), kl_c.data_ptr(), kh_c.data_ptr(), local_rank, stream), "lf_ntt")
        for _ in range(5):
            step_ct()
        torch.cuda.synchronize()
        ct_ms = event_time_ms(step_ct, max(10, args.steps // 2))
        extra["poly_ntt_per_s_ciphertext_limbs_1_integer_class"] = B / (ct_ms * 1e-3)
        extra["ciphertext_limbs_note"] = (f"rows {lo_ct}..{total - 5}: 29 scale primes + base prime (a level-5 ciphertext, no special primes);"
                                          " reported for the limb-mix sensitivity only, the metric above keeps its 5 integer-class limbs")
        rates, roof = engine_rates(dev, quick=False)
        extra.update(rates)
        result["roofline_engine_ops"] = roof     # cc_mult_evk / rotate_single: the metric's second half, per preset
        result["cpu_baseline"] = cpu_baseline(ctx, rows_idx, batch=min(B, 16))   # bounded sample of the same workload
        for preset, budget, reps in (("silver", 20.0, 10), ("bronze", 6.0, 10), ("gold", 15.0, 3)):
            try:
                result["cpu_baseline"][f"cc_mult_evk_{preset}"] = cpu_engine_baseline(preset, budget, reps)
            except Exception as e:
                result["cpu_baseline"][f"cc_mult_evk_{preset}"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        for preset in ("bronze", "silver", "gold"):     # the oracle transform at every preset's full chain
            try:
                result["cpu_baseline"][f"ntt_{preset}"] = cpu_ntt_baseline_preset(preset)
            except Exception as e:
                result["cpu_baseline"][f"ntt_{preset}"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    else:
        result["cpu_baseline"] = None   # reported by the N=1 run only
    if world > 1 and not args.no_extra:
        # The engine legs run under a watchdog: if they are not back after --leg-timeout seconds (a rank stuck in
        # an exchange), rank 0 prints the line with what has been measured so far and every rank leaves with
        # os._exit(3) — nothing is re-exec'ed, the process just ends, and the launcher reports the failure.
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(args.leg_timeout):
                extra["multi_gpu_legs_abandoned_after_s"] = args.leg_timeout
                if rank == 0:
                    print(json.dumps(result), flush=True)
                os._exit(3)     # a hung leg is a failure: the partial line is printed, the exit code says so
        threading.Thread(target=watchdog, daemon=True).start()
        multi_gpu_rates(dev, world, rank, extra, sharded=not args.no_sharded)
        done.set()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
