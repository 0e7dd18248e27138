#!/usr/bin/env python3
"""Headline benchmark: NTTs/sec/GPU at logN=16, L=30 limbs (+ cc_mult_evk ops/sec), % of HBM roofline.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

One "step" = one forward negacyclic NTT (ntt_cuda.ntt semantics, bit-exact lazy outputs) of a batch of
B polynomials x 30 RNS limbs at logN = 16 — the gold preset's with-special row set at level 9 (25 scale
primes + base + 4 special primes) — through the C ABI entry lf_ntt_ws: lf_ntt with a resident workspace of the operands' size
between its two launches, what ntt_cuda.ntt does for a caller (LF_BENCH_NTT_INPLACE=1: lf_ntt; extra.poly_ntt_per_s_in_place_lf_ntt
reports that form beside the metric).  B = 128 by default (3.75 GiB-class working set: B x 30 MiB, far beyond the
256 MiB Infinity Cache; at B = 16 the column pass still finds part of its input there and the launch tail of the
tiled pass is 10 block rounds instead of 80).  Inputs are synthetic (splitmix64 -> mod 2q), resident in HBM before the timed region.
Polynomials are independent, so ranks shard them with no data-path collective: weak scaling.

The JSON line also carries
  roofline     : the dominant kernel (ntt_pass16_fwd_seq_ws: the tiled pass of all 30 limbs, the second of the two
                 launches of a transform, 11 of its 16 stages) against the 8 TB/s HBM peak; algorithmic bytes
                 per launch = 8*N*limbs (a transform is 16*N bytes per limb, SURVEY.md §8d, spread over its two
                 launches); its launch duration is measured live with HIP events on the launch stream, the
                 kernel launched alone (lf_ntt_pass_ws, the measurement entry) with the grid it has inside the full step;
                 `shader_clock_mhz` = the clock the CUs run at beside each kernel (lf_clock_probe on a second stream) and
                 `package_power` = the amdgpu hwmon reading while the dominant kernel runs: the tiled pass runs at ~1.93 of
                 2.4 GHz with its VALUs 91 % busy — `frac` is what that clock allows;
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
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); it has to be in
# the environment before the HIP runtime starts, i.e. before anything below touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def physical_cores():
    """One logical CPU per physical core this process may run on (SMT siblings counted once), in CPU order."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out or [0]


def cpu_quota_cores():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unreadable:
    more runnable threads than this only take turns."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(ctx, rows_idx, batch_cap=64, budget_s=12.0):
    """The oracle's NTT on the host cores over a [batch*30, N] stack of the same workload.  Threads = min(rows, physical
    cores), each pinned to its own core (oracle.pin_threads), static schedule, and every row — data and twiddles — first
    touched by the thread that transforms it (oracle.place_rows), so nothing is read across sockets; the batch is the
    smallest one <= batch_cap whose rows divide evenly over the threads (else batch_cap), so no thread idles in the last
    round.  `value` = this threaded leg; `single_thread` = the same kernel on ONE core, the number a GPU/CPU ratio
    should be read against."""
    from oracle import oracle as orc
    L = len(rows_idx)
    cpus = physical_cores()
    visible = len(cpus)
    quota = cpu_quota_cores()
    if quota is not None and quota < len(cpus):     # a container limited to fewer CPU-seconds per second than it can see cores
        cpus = cpus[:max(1, int(quota))]
    cores = len(cpus)
    batch = next((b for b in range(1, batch_cap + 1) if b * L >= cores and (b * L) % cores == 0), batch_cap)
    n = L * batch
    threads = min(cores, n)
    before = orc.omp_threads(threads)
    affinity = os.sched_getaffinity(0)
    unpinned = orc.pin_threads(cpus[:threads])
    try:
        col = lambda v: np.asarray([v[i] for i in rows_idx], dtype=np.int64)
        h = lambda v: np.ascontiguousarray(np.tile(col(v), batch))
        ql, qh, kl, kh = h(ctx.q_lower_bits), h(ctx.q_higher_bits), h(ctx.k_lower_bits), h(ctx.k_higher_bits)
        q2 = h(ctx.q_double)
        psi1 = np.ascontiguousarray(ctx.psi_br[rows_idx].copy())
        orc.mont_enter(psi1, col(ctx.R_square), L, ql[:L], qh[:L], kl[:L], kh[:L])
        psi = orc.place_rows(psi1, n)                       # row r's twiddles beside row r's thread
        rng = np.random.default_rng(5)
        x = orc.place_rows(rng.integers(0, 1 << 40, size=(L, ctx.N), dtype=np.int64), n)
        def threaded(budget):
            orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)   # warm (thread pool)
            t0, reps = time.time(), 0
            while True:
                orc.ntt(x, psi, n, ctx.logN, q2, ql, qh, kl, kh)
                reps += 1
                if time.time() - t0 > budget or reps >= 200:
                    break
            return (time.time() - t0) / reps, reps
        # two builds of the same C text (oracle/oracle.py): the portable -O2 one every test uses, and -O3 -march=native compiled
        # on THIS box — the fair port; `value` is the faster of the two, both are in the line
        builds = {}
        dt_p, reps_p = threaded(0.4 * budget_s)
        builds["portable"] = {"value": batch / dt_p, "flags": orc.build_flags("portable"), "reps": reps_p}
        try:
            orc.use_build("native")
            dt_n, reps_n = threaded(0.6 * budget_s)
            builds["native"] = {"value": batch / dt_n, "flags": orc.build_flags("native"), "reps": reps_n}
        except Exception as e:
            builds["native"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            orc.use_build("portable")
        best = max((k for k in builds if "value" in builds[k]), key=lambda k: builds[k]["value"])
        orc.use_build(best)
        dt, reps = batch / builds[best]["value"], builds[best]["reps"]
        out = {"value": batch / dt, "unit": "poly-NTT(L=30,logN=16)/s", "cores": threads, "kind": "port", "build": builds[best]["flags"],
               "builds": builds,
               "sample": f"{reps} x forward NTT of {batch} polys x {L} limbs, N=65536, C oracle ({builds[best]['flags']}), OpenMP static over limb rows, "
                         f"{threads} threads pinned one per physical core ({cores} usable: {visible} visible, "
                         f"cgroup CPU quota {quota if quota is not None else 'none'}; {unpinned} not pinned), "
                         "rows and twiddles first-touched by their thread"}
        # the same kernel on ONE core: one polynomial's 30 limbs, a single OpenMP thread
        orc.omp_threads(1)
        x1 = np.ascontiguousarray(x[:L])
        orc.ntt(x1, psi[:L], L, ctx.logN, q2[:L], ql[:L], qh[:L], kl[:L], kh[:L])
        t0, reps1 = time.time(), 0
        while True:
            orc.ntt(x1, psi[:L], L, ctx.logN, q2[:L], ql[:L], qh[:L], kl[:L], kh[:L])
            reps1 += 1
            if time.time() - t0 > 4.0 or reps1 >= 20:
                break
        out["single_thread"] = {"value": reps1 / (time.time() - t0), "unit": "poly-NTT(L=30,logN=16)/s", "cores": 1,
                                "sample": f"{reps1} x forward NTT of 1 poly x {L} limbs, one OpenMP thread"}
        out["threads_speedup_over_one"] = out["value"] / out["single_thread"]["value"]
        return out
    finally:
        orc.use_build("portable")             # the checker build for everything after this leg
        orc.omp_threads(before)
        os.sched_setaffinity(0, affinity)     # thread 0 of the team is this thread: give it its CPUs back


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


def classify_bound(pmc_kernels):
    """What bounds an engine op, from the per-kernel PMC table of THIS build (profiles/<tag>_engine_ops_pmc.json):
    a kernel counts as `hbm` when it moves >= 70 % of the 6.3 TB/s a streaming kernel reaches, as `valu_issue` when its
    SIMDs issue a VALU instruction in >= 70 % of the busy cycles, else as `latency` (neither pipe saturated: short
    launches, dependent LDS exchanges); the op's bound is the class holding most of its kernel time.  Returns
    (bound, {class: share of the op's kernel time}) or (None, None) without counters."""
    if not pmc_kernels:
        return None, None
    share = {"hbm": 0.0, "valu_issue": 0.0, "latency": 0.0}
    for k in pmc_kernels:
        us = k["us"] * k.get("per_op", 1)
        cls = "hbm" if k["moved_TBps"] >= 0.7 * HBM_ACHIEVABLE_GBS / 1e3 else "valu_issue" if k["valu_busy"] >= 0.70 else "latency"
        share[cls] += us
    total = sum(share.values()) or 1.0
    share = {k: round(v / total, 3) for k, v in share.items()}
    return max(share, key=share.get), share


def op_roofline(eng, op, ops_per_s, profile, pmc=None):
    rows = algorithmic_rows(eng, op)
    nbytes = rows * eng.ctx.N * 8
    achieved = nbytes * ops_per_s / 1e9
    bound, share = classify_bound((pmc or {}).get("kernels"))
    out = {"bound": bound, "bound_shares": share, "algorithmic_rows": rows, "algorithmic_bytes": nbytes, "ops_per_s": ops_per_s,
           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "frac_of_achievable_6300": achieved / HBM_ACHIEVABLE_GBS}
    if bound is None:
        out["bound_note"] = "no PMC table of this build (profiles/<tag>_engine_ops_pmc.json, digest-stamped): bound not stated"
    if pmc:
        out["traffic"] = pmc.get("bytes_per_op")      # HBM bytes per op from the PMC passes (reads corrected, MI355X_MICROARCH.md)
        out["traffic_over_algorithmic"] = None if not pmc.get("bytes_per_op") else pmc["bytes_per_op"] / nbytes
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
    pmc = stamped_profile("r*_engine_ops_pmc.json") or {}        # per-kernel HBM bytes / VALU utilisation, same rule
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
        roof[f"cc_mult_evk_{name}"] = op_roofline(eng, "cc_mult", 1e3 / ms, prof.get(f"{name}_cc_mult"), pmc.get(f"{name}_cc_mult"))
        ms = event_time_ms(lambda: eng.rotate_single(a, rotk), n)
        out[f"rotate_single_{name}_ops_per_s"] = 1e3 / ms
        roof[f"rotate_single_{name}"] = op_roofline(eng, "rotate", 1e3 / ms, prof.get(f"{name}_rotate"), pmc.get(f"{name}_rotate"))
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
        roof[f"cc_mult_evk_{name}_batch{nb}"] = op_roofline(eng, "cc_mult", nb * 1e3 / ms, None, pmc.get(f"{name}_cc_mult_batch"))
        if name == "gold":
            # BASELINE configs[4] at its stated size: 64 level-0 ciphertexts (seeds 100..163) under one rotation key
            cts64 = cts + [synth.ciphertext(eng, 100 + i, 0) for i in range(nb, 64)]
            eng.rotate_single_batch(cts64, rotk)
            torch.cuda.synchronize()
            ms = event_time_ms(lambda: eng.rotate_single_batch(cts64, rotk), 3)
            out["rotate_single_gold_batch64_rotations_per_s"] = 64 * 1e3 / ms
            roof["rotate_single_gold_batch64"] = op_roofline(eng, "rotate", 64 * 1e3 / ms, None, pmc.get("gold_rotate_batch"))
            del cts64
        del cts, pairs
        del eng, a, b, evk, rotk
        torch.cuda.empty_cache()
    return out, roof


def ntt_rates_preset(name, dev, batch, iters=20):
    """BASELINE's metric is quoted at logN 15 AND 16: forward (lf_ntt_ws = ntt_cuda.ntt, exact lazy words; logN 17: lf_ntt) and inverse (lf_intt
    with tail 2 = ntt_cuda.intt_exit_reduce) transforms of `batch` polynomials x EVERY limb of the preset's chain (special
    primes included: silver 19, gold 39, platinum 59+) through the C ABI, inputs resident in HBM, each with its roofline
    block (16 N bytes per limb-NTT).  Outside the headline's timed region; N = 1 only."""
    from liberate_fhe_amd._native import lib, check
    from liberate_fhe_amd.fhe import presets
    from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
    from liberate_fhe_amd.ntt import ntt_context, twiddles
    from liberate_fhe_amd.utils import synth
    params = {k: v for k, v in presets.params[name].items() if k != "devices"}
    ctx = ckks_context(**params)
    ntt = ntt_context(ctx, devices=[dev])
    L, N, logN = len(ctx.q), ctx.N, ctx.logN
    d = torch.device(dev).index or 0
    x = torch.empty((batch, L, N), dtype=torch.int64, device=dev)
    one = torch.from_numpy(synth.uniform_rows(77, list(range(L)), ctx.q, N, lazy=True)).to(dev)
    x[:] = one
    psi, ipsi, q2, ql, qh, kl, kh, ninv = (t[0] for t in (ntt.psi, ntt.ipsi, ntt._2q, ntt.ql, ntt.qh, ntt.kl, ntt.kh, ntt.Ninv))
    st = torch.cuda.current_stream().cuda_stream
    dp, idp = twiddles.dp_pointer(psi, ql, qh, kl, kh, d, st), twiddles.dp_pointer(ipsi, ql, qh, kl, kh, d, st)
    q_host = np.array(ctx.q, dtype=np.int64)

    ws = torch.empty((int(lib.lf_ntt_ws_words(batch, L, logN)),), dtype=torch.int64, device=dev)   # as ntt_cuda.ntt keeps one

    def fwd():
        check(lib.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), batch, L, logN, psi.data_ptr(), dp, q_host.ctypes.data, 0, 0, ql.data_ptr(),
                            qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), d, st), "lf_ntt_ws")

    def inv():
        check(lib.lf_intt_ws(x.data_ptr(), ws.data_ptr(), batch, L, logN, ipsi.data_ptr(), idp, q_host.ctypes.data, ninv.data_ptr(), 2, 0,
                             ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), d, st), "lf_intt_ws")

    def inv_in_place():
        check(lib.lf_intt(x.data_ptr(), batch, L, logN, ipsi.data_ptr(), idp, q_host.ctypes.data, ninv.data_ptr(), 2, 0, q2.data_ptr(),
                          ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), d, st), "lf_intt")

    # round trip first: intt_exit_reduce(enter_ntt(x)) is x's canonical residues (nctx.py:554-579; the plain `ntt` timed below
    # skips the Montgomery entry, so its inverse chain would return x R^-1)
    rs = ntt.Rs[0]
    check(lib.lf_ntt(x.data_ptr(), batch, L, logN, psi.data_ptr(), dp, q_host.ctypes.data, rs.data_ptr(), 0, q2.data_ptr(), ql.data_ptr(),
                     qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), d, st), "lf_ntt (enter_ntt)")
    inv()
    torch.cuda.synchronize()
    qcol = torch.tensor(ctx.q, dtype=torch.int64, device=dev)[:, None]
    ok = bool(torch.equal(x[0], one % qcol))
    out = {"limbs": L, "logN": logN, "batch": batch, "round_trip_identity": ok}
    nbytes = 16 * N * L * batch
    for key, fn, prep in (("forward", fwd, None), ("inverse_exit_reduce", inv, fwd), ("inverse_exit_reduce_in_place_lf_intt", inv_in_place, fwd)):
        x[:] = one
        if prep:
            prep()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = event_time_ms(fn, iters)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[key] = {"poly_ntt_per_s": batch / (ms * 1e-3), "limb_ntt_per_s": batch * L / (ms * 1e-3), "ms_per_step": ms,
                    "roofline": {"bound": "valu_issue" if logN >= 13 else "latency", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_step": nbytes,
                                 "note": "whole transform (both launches), 16 N bytes per limb; no PMC table for this shape"}}
    del x, ws
    torch.cuda.empty_cache()
    return out


def platinum_rates(dev):
    """The reference's largest preset (logN 17, 6 special primes; presets/params.py): cc_mult(+relinearize) and rotate_single
    ops/s.  logN 17 takes the 8-words tiled kernels and the LDS-tiled extension."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    from liberate_fhe_amd.utils import synth
    eng = ckks_engine(**{**presets.params["platinum"], "devices": [dev]})
    a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
    evk = synth.key_switch_key(eng, 5)
    rotk = synth.key_switch_key(eng, 6, origin="rotation key:1")
    for _ in range(10):
        eng.cc_mult(a, b, evk)
        eng.rotate_single(a, rotk)
    torch.cuda.synchronize()
    out = {}
    ms = event_time_ms(lambda: eng.cc_mult(a, b, evk), 20)
    out["cc_mult_evk_platinum_ops_per_s"] = 1e3 / ms
    out["roofline_cc_mult"] = op_roofline(eng, "cc_mult", 1e3 / ms, None)
    ms = event_time_ms(lambda: eng.rotate_single(a, rotk), 20)
    out["rotate_single_platinum_ops_per_s"] = 1e3 / ms
    out["roofline_rotate"] = op_roofline(eng, "rotate", 1e3 / ms, None)
    out["limbs_level0_with_special"] = len(eng.ntt.p.destination_arrays_with_special[0][0])
    del eng, a, b, evk, rotk
    torch.cuda.empty_cache()
    return out


def api_endpoints(dev):
    """Wall-clock of the API endpoints the reference's example notebook times (`examples/[Example] CKKS
    engine.ipynb` cells 6-12, silver, unnamed NVIDIA GPU — context, not a target).  Every figure here is SYNCHRONISED
    (the notebook's key timings are asynchronous launch times); `first_call` includes table builds, `warm` is the median
    of 20 calls after it."""
    from liberate_fhe_amd.fhe import ckks_engine, presets
    out = {"preset": "silver", "reference_notebook": {"engine_construction_s": 2.63, "create_secret_key_us": 873, "create_public_key_us": 633,
                                                      "encode_encrypt_decrypt_decode_first_call_ms": 99, "encorypt_decrode_ms": 4.02,
                                                      "note": "unnamed NVIDIA GPU; key timings there are asynchronous launches, no sync"}}

    def timed(fn, reps=20):
        ts = []
        res = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], res

    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng = ckks_engine(**{**presets.params["silver"], "devices": [dev]})
        torch.cuda.synchronize()
        out["engine_construction_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng2 = ckks_engine(**{**presets.params["silver"], "devices": [dev]})
        torch.cuda.synchronize()
        out["engine_construction_second_s"] = time.perf_counter() - t0
        del eng2
        t0 = time.perf_counter()
        sk = eng.create_secret_key()
        torch.cuda.synchronize()
        out["create_secret_key_first_call_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        pk = eng.create_public_key(sk)
        torch.cuda.synchronize()
        out["create_public_key_first_call_ms"] = 1e3 * (time.perf_counter() - t0)
        dt, sk = timed(eng.create_secret_key)
        out["create_secret_key_ms"] = 1e3 * dt
        dt, pk = timed(lambda: eng.create_public_key(sk))
        out["create_public_key_ms"] = 1e3 * dt
        dt, evk = timed(lambda: eng.create_evk(sk), 5)
        out["create_evk_ms"] = 1e3 * dt
        m = eng.example(-1, 1)
        # the first FFT of a process builds its rocFFT plan (seconds on this image): timed on its own so that the first
        # encorypt + decrode below shows what is ours (table builds) and not the library's one-time set-up
        t0 = time.perf_counter()
        torch.fft.fft(torch.zeros(eng.ctx.N, dtype=torch.complex128, device=dev))
        torch.cuda.synchronize()
        out["torch_fft_first_call_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        back = eng.decrode(eng.encorypt(m, pk), sk)
        out["encorypt_decrode_first_call_ms"] = 1e3 * (time.perf_counter() - t0)
        out["encorypt_decrode_max_abs_error"] = float(np.abs(back - m).max())
        dt, ct = timed(lambda: eng.encorypt(m, pk))
        out["encorypt_ms"] = 1e3 * dt
        dt, _ = timed(lambda: eng.decrode(ct, sk))
        out["decrode_ms"] = 1e3 * dt
        dt, _ = timed(lambda: eng.decrode(eng.encorypt(m, pk), sk))
        out["encorypt_decrode_ms"] = 1e3 * dt
        # the mult endpoint on top of it (what the metric's second half times, here through the public dispatch)
        ct2 = eng.encorypt(m, pk)
        dt, prod = timed(lambda: eng.mult(ct, ct2, evk))
        out["mult_ms"] = 1e3 * dt
        out["mult_decode_max_abs_error"] = float(np.abs(eng.decrode(prod, sk) - m * m).max())
        del eng
        torch.cuda.empty_cache()
    except Exception as e:   # a context figure, never a reason to lose the line
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


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


def _reduce_device(dev):
    """Where the tensors of the DEFAULT group's reductions live: the host — the default group is gloo on every run
    (main(): RCCL is a second group, agreed on collectively)."""
    return "cpu"


def _max_over_ranks(ms, dev):
    import torch.distributed as dist
    t = torch.tensor([ms], dtype=torch.float64, device=_reduce_device(dev))
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


def comm_prepare(dev, world, rank, local_rank, grp=None):
    """N > 1, BEFORE the watchdog starts: the process group of the limb-sharded legs and one untimed all-pairs exchange on
    it (`grp` = the RCCL group main() brought up and the ranks agreed on; None in the rehearsal: a gloo group).  RCCL sets a point-to-point channel up lazily, on a pair's first message; the first key switch of the sharded
    engine addresses all N (N - 1) directed pairs in one group — that set-up belongs here, under the communicator's
    long timeout, not inside the 300 s watchdog of the timed legs.  Returns (group, comm block of the JSON line)."""
    import datetime
    import torch.distributed as dist
    # (its timeout must NOT undercut the watchdog: a communicator that times out first tears the process down before the
    # line is printed)
    if grp is None:    # rehearsal: gloo + host staging
        grp = dist.new_group(ranks=list(range(world)), timeout=datetime.timedelta(seconds=900))
    pr = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "device": dev, "name": pr.name,
          "pci": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}.0",
          "uuid": str(getattr(pr, "uuid", ""))}
    seen = [None] * world
    dist.all_gather_object(seen, me, group=grp)
    t0 = time.perf_counter()
    box = torch.full((world, 64), rank, dtype=torch.int64, device=dev)
    ops = []
    for p in range(world):
        if p != rank:
            ops.append(dist.P2POp(dist.isend, box[rank], dist.get_global_rank(grp, p), grp))
            ops.append(dist.P2POp(dist.irecv, box[p], dist.get_global_rank(grp, p), grp))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    ok = bool((box == torch.arange(world, device=dev, dtype=torch.int64)[:, None]).all().item())
    block = {"backend": str(dist.get_backend(grp)), "world_size": dist.get_world_size(grp), "ranks_seen": len([x for x in seen if x]),
             "devices": seen, "distinct_pci_ids": len({x["pci"] for x in seen if x}),
             "p2p_warmup_s": time.perf_counter() - t0, "p2p_warmup_all_pairs_ok": ok,
             "transport": "RCCL point-to-point batches (ncclGroup of send / recv) over xGMI" if "nccl" in str(dist.get_backend(grp))
                          else "gloo + host staging (REHEARSAL transport, tests/gloo_device_p2p.py)"}
    block["selftest"] = comm_selftest(grp, dev, world, rank)
    return grp, block


def comm_selftest(grp, dev, world, rank):
    """The communication PATTERNS of the limb-sharded engine once each on small device buffers, with checked contents, before
    any engine runs — so that a first contact with RCCL that goes wrong is attributed to a pattern, not to "cc_mult hung":
      fanout      one owner sends one buffer to every other rank in ONE batch (the rescale row, comm.fanout_into);
      unequal     an all-pairs batch whose messages differ in size per owner (runs of digits: 7 rows from rank 0, 4 from the rest);
      subset      the same among the first world - 1 ranks only, the last rank sitting it out (levels where a rank has no rows);
      allgather   one padded all-gather into a flat tensor (DistComm(exchange="allgather")).
    Every rank learns every verdict (a MIN over ranks on the default group)."""
    import torch.distributed as dist
    from liberate_fhe_amd.fhe.comm import DistComm
    out = {}
    comm = DistComm(group=grp, local_device=dev)
    W = 256

    def verdict(name, fn):
        t0 = time.perf_counter()
        try:
            ok = bool(fn())
            err = None
        except Exception as e:
            ok, err = False, f"{type(e).__name__}: {e}"[:200]
        torch.cuda.synchronize()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        out[name] = {"ok_on_every_rank": bool(int(flag.item())), "ms": round(1e3 * (time.perf_counter() - t0), 2)}
        if err:
            out[name]["error_on_this_rank"] = err

    def fanout():
        buf = torch.full((4, W), 7 if rank == 0 else -1, dtype=torch.int64, device=dev)
        comm.fanout_into(buf, 0, list(range(world)))
        torch.cuda.synchronize()
        return (buf == 7).all().item()

    def exchange(peers, form):
        rows = [7 if r == 0 else 4 for r in range(len(peers))]
        starts = [sum(rows[:i]) for i in range(len(peers))]
        buf = torch.full((sum(rows), W), -1, dtype=torch.int64, device=dev)
        pieces = [(r, starts[i], rows[i]) for i, r in enumerate(peers)]
        if rank in peers:
            i = peers.index(rank)
            buf[starts[i]:starts[i] + rows[i]] = 100 + rank
        c = comm if form == "p2p" else DistComm(group=grp, local_device=dev, exchange="allgather")
        if rank not in peers and form == "p2p":
            return True                                    # takes no part in the point-to-point form
        c.exchange_rows(buf if rank in peers else None, pieces, peers, width=W).wait()
        torch.cuda.synchronize()
        if rank not in peers:
            return True
        want = torch.cat([torch.full((rows[i], W), 100 + r, dtype=torch.int64, device=dev) for i, r in enumerate(peers)])
        return torch.equal(buf, want)

    verdict("fanout", fanout)
    verdict("unequal_all_pairs", lambda: exchange(list(range(world)), "p2p"))
    if world > 2:
        verdict("subset_without_last_rank", lambda: exchange(list(range(world - 1)), "p2p"))
    verdict("allgather_padded", lambda: exchange(list(range(world)), "allgather"))
    if world > 2:
        verdict("allgather_with_idle_rank", lambda: exchange(list(range(world - 1)), "allgather"))
    return out


def link_bytes(eng, level):
    """Bytes one key switch at `level` puts on each directed link (owner -> peer) of the point-to-point batch, and the
    rows each rank owns: from the engine's exchange schedule (`_ks_tables(level)["groups"]`)."""
    tabs = eng._ks_tables(level)
    n_alive = eng.len_devices[level]
    per_owner = {}
    for owner, first, count, row0, nrows, src_row in tabs["groups"]:
        per_owner[owner] = per_owner.get(owner, 0) + nrows * eng.ctx.N * 8
    return {"alive_ranks": n_alive, "bytes_per_link_by_owner": per_owner,
            "max_bytes_per_link": max(per_owner.values()) if per_owner else 0,
            "bytes_received_per_rank": {r: sum(v for o, v in per_owner.items() if o != r) for r in range(n_alive)},
            "messages_per_batch_per_rank": {r: sum((n_alive - 1) if g[0] == r else 1 for g in tabs["groups"]) for r in range(n_alive)}}


def host_enqueue_us(fn, bursts=7, n=8):
    """Median host time to ENQUEUE one op (bursts of `n` from an idle queue, no synchronisation inside a burst)."""
    ts = []
    for _ in range(bursts):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ts.append((time.perf_counter() - t0) / n)
        torch.cuda.synchronize()
    ts.sort()
    return 1e6 * ts[len(ts) // 2]


def multi_gpu_rates(dev, world, rank, out, sharded=True, grp=None, comm_block=None):
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
    from liberate_fhe_amd.fhe.comm import DistComm
    try:
        if grp is None:
            grp, comm_block = comm_prepare(dev, world, rank, torch.cuda.current_device())
    except Exception as e:
        out["multi_gpu_limb_sharded_error"] = f"{type(e).__name__}: {e}"[:300]
        return
    # both forms of the key-switch digit exchange in one run, the parity gate in front of each: one batch of point-to-point
    # messages between the ranks that hold rows (the default), and ONE padded all-gather over the group
    for mode, suffix in (("p2p", ""), ("allgather", "_allgather")):
        try:
            eng = ckks_engine(devices=[dev], comm=DistComm(group=grp, local_device=dev, exchange=mode), **params)
            if comm_block is not None and mode == "p2p":
                comm_block["key_switch_batch_level0"] = link_bytes(eng, 0)
                comm_block["key_switch_batch_level10"] = link_bytes(eng, 10)
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
                out[f"multi_gpu_limb_sharded{suffix}_error"] = (f"parity gate: {int(t.item())} row(s) of the limb-sharded results differ from the "
                                                               "unsharded engine (or no reference was computed); rates withheld")
                continue
            out[f"limb_sharded{suffix}_parity"] = ("bit-exact vs the unsharded engine on every rank: cc_mult 0->1, rotate at 0, cc_mult 9->10, "
                                                  "two rotations at level 10 (at 8 ranks the last one holds no rows there)")
            a, b = synth.ciphertext(eng, 3, 0), synth.ciphertext(eng, 4, 0)
            for _ in range(20):
                eng.cc_mult(a, b, evk)
            torch.cuda.synchronize()
            dist.barrier(group=grp)
            ms = _max_over_ranks(event_time_ms(lambda: eng.cc_mult(a, b, evk), 40), dev)
            out[f"cc_mult_evk_gold_limb_sharded{suffix}_ops_per_s"] = 1e3 / ms
            for _ in range(10):
                eng.rotate_single(a, rotk)
            torch.cuda.synchronize()
            dist.barrier(group=grp)
            ms = _max_over_ranks(event_time_ms(lambda: eng.rotate_single(a, rotk), 40), dev)
            out[f"rotate_single_gold_limb_sharded{suffix}_ops_per_s"] = 1e3 / ms
            out["limb_sharded_rows_per_rank_level0"] = [len(d) for d in eng.ntt.p.destination_arrays_with_special[0]]
            out[f"limb_sharded{suffix}_hip_graphs"] = bool(eng.graph_sharded)
            # host time this rank spends ENQUEUEING one sharded op (the margin before the host, not the GPU, paces a rank)
            if comm_block is not None:
                mine = torch.tensor([host_enqueue_us(lambda: eng.cc_mult(a, b, evk)), host_enqueue_us(lambda: eng.rotate_single(a, rotk))],
                                    dtype=torch.float64, device=dev)
                every = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(every, mine, group=grp)
                comm_block[f"host_enqueue_us_per_sharded_cc_mult_by_rank{suffix}"] = [round(float(t[0]), 1) for t in every]
                comm_block[f"host_enqueue_us_per_sharded_rotate_by_rank{suffix}"] = [round(float(t[1]), 1) for t in every]
            del eng, evk, rotk, a, b
            torch.cuda.empty_cache()
        except Exception as e:   # the headline line must survive a failure of this leg
            out[f"multi_gpu_limb_sharded{suffix}_error"] = f"{type(e).__name__}: {e}"[:300]


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


def run_rccl_world1(timeout=420):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    script = os.path.join(ROOT, "tools", "rccl_world1.py")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        block = json.loads(lines[-1]) if lines else {"ok": False, "error": "no JSON line", "stderr_tail": r.stderr[-400:]}
        block["exit_code"] = r.returncode
    except Exception as e:
        block = {"ok": False, "error": f"{type(e).__name__}: {e}"[:300]}
    block["wall_s"] = round(time.perf_counter() - t0, 1)
    return block


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
    # N = 1: the only RCCL evidence a one-GPU lease allows — tools/rccl_world1.py as a CHILD process, started and finished before
    # this process touches the GPU (nothing is exec'ed over a GPU process, nothing runs beside the timed region): a real
    # communicator of one rank under the engine's sharded code path, both exchange forms, graphs on and off, the stream-ordering
    # semantics of work.wait(); its JSON becomes comm.rccl_world1 of the line.
    rccl_world1 = None
    if world == 1 and not args.no_extra and os.environ.get("LF_BENCH_RCCL_WORLD1", "1") != "0" and torch.cuda.device_count() >= 1:
        import __graft_entry__ as g0
        g0.build()                      # (hipcc cross-compiles without touching the GPU: the child must load THIS tree's library)
        rccl_world1 = run_rccl_world1()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # development aid (never set by the driver): LF_BENCH_REHEARSE=1 runs the N > 1 code path on a ONE-GPU box —
    # every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one device).  Numbers are meaningless.
    rehearse = world > 1 and os.environ.get("LF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    nccl_error = None
    rccl = None          # the RCCL process group (barriers of the timed region, the limb-sharded legs); None: not up
    if world > 1:
        import datetime
        import torch.distributed as dist
        # The DEFAULT group is gloo on every run: it always comes up (host TCP), carries the host-side reductions (max over
        # ranks, parity counters) and is what the ranks use to AGREE on whether RCCL is usable — a rank-local try/except
        # around an RCCL initialisation would leave mixed backends behind a partial failure, and the job would hang.
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
        if rehearse:
            from tests import gloo_device_p2p      # gloo moves host memory only: device messages are staged (test transport)
            gloo_device_p2p.install()
        else:
            try:
                rccl = dist.new_group(ranks=list(range(world)), backend="nccl", timeout=datetime.timedelta(seconds=900))
                probe = torch.ones(1, dtype=torch.int64, device=dev)
                dist.all_reduce(probe, group=rccl)             # first RCCL collective of the process: communicator set-up
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"RCCL all_reduce of ones returned {int(probe.item())}, expected {world}")
            except Exception as e:
                nccl_error = f"{type(e).__name__}: {e}"[:300]
                print(f"[bench] rank {rank}: RCCL did not come up ({nccl_error})", file=sys.stderr, flush=True)
            flag = torch.tensor([0 if nccl_error is None else 1], dtype=torch.int64)
            dist.all_reduce(flag)                              # gloo: every rank learns the same answer
            if int(flag.item()):
                if nccl_error is None:
                    nccl_error = f"RCCL came up here but failed on {int(flag.item())} of {world} ranks"
                rccl = None

    def barrier():
        """Barrier of the timed region: on RCCL (a device-side collective on the compute device) when it is up, and always on
        the default gloo group."""
        if world > 1:
            if rccl is not None:
                dist.barrier(group=rccl)
            dist.barrier()

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    barrier()

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

    # The transform goes through a workspace (lf_ntt_ws, include/ckks_hip.h: the two passes exchange the fp64-class limbs as
    # 6-byte planes instead of 8-byte words; same result words on any input) — what ntt_cuda.ntt does for a caller; the
    # workspace is resident like the operands.  LF_BENCH_NTT_INPLACE=1: the strictly in-place lf_ntt instead (experiments).
    in_place = os.environ.get("LF_BENCH_NTT_INPLACE", "0") == "1" or ntt_flags != 0
    ws = None if in_place else torch.empty((int(lib.lf_ntt_ws_words(B, L_LIMBS, LOGN)),), dtype=torch.int64, device=dev)

    def step_in_place():
        check(lib.lf_ntt(x.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, ntt_flags, q2.data_ptr(), ql.data_ptr(),
                         qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt")

    def step_ws():
        check(lib.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, ql.data_ptr(),
                            qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt_ws")

    step = step_in_place if in_place else step_ws

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / args.steps
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=_reduce_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall / args.steps * 1e3

    value = world * B / (wall / args.steps)                         # poly-NTT(30)/s, whole job

    # Parity spot check of what the timed kernel computes, on THIS box, outside the timed region: polynomial 0 is reloaded
    # from its seed, the whole batch goes through one more step, and two of its limbs — the first integer-class one (a
    # 60-bit prime, REDC62 path) and the first fp64-class one (a 40-bit prime) — are compared word for word with the C oracle.
    # The WHOLE batch of that step is also compared, on the device, with the strictly in-place lf_ntt of the same input
    # (tests/test_fullsize_gpu.py holds both forms to the oracle at this shape, all 30 limbs, out-of-range words included).
    def parity_spot_check():
        from oracle import oracle as orc      # the checker, never the thing measured
        src = synth.uniform_rows(1000 * rank, rows_idx, ctx.q, N, lazy=True)
        x[0] = torch.from_numpy(src).to(dev)
        before = None if in_place else x.clone()     # the whole batch again through the strictly in-place lf_ntt, on the device
        step()
        torch.cuda.synchronize()
        if before is not None:
            check(lib.lf_ntt(before.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, q2.data_ptr(), ql.data_ptr(),
                             qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt")
            torch.cuda.synchronize()
            same = torch.equal(before, x)
            n_diff = 0 if same else int((before != x).sum().item())
            del before
            if not same:
                return f"MISMATCH: {n_diff} words of the whole batch differ between lf_ntt_ws and lf_ntt"
        got = x[0].cpu().numpy()
        h = lambda v, i: np.asarray([v[i]], dtype=np.int64)
        picks = [next(r for r, i in enumerate(rows_idx) if ctx.q[i] >= (1 << 41)), next(r for r, i in enumerate(rows_idx) if ctx.q[i] < (1 << 41))]
        for r in picks:
            i = rows_idx[r]
            ql_, qh_, kl_, kh_ = h(ctx.q_lower_bits, i), h(ctx.q_higher_bits, i), h(ctx.k_lower_bits, i), h(ctx.k_higher_bits, i)
            psi_row = np.ascontiguousarray(ctx.psi_br[i:i + 1].copy())
            orc.mont_enter(psi_row, h(ctx.R_square, i), 1, ql_, qh_, kl_, kh_)
            want = np.ascontiguousarray(src[r:r + 1].copy())
            orc.ntt(want, psi_row, 1, ctx.logN, h(ctx.q_double, i), ql_, qh_, kl_, kh_)
            if not (want[0] == got[r]).all():
                return f"MISMATCH: limb {r} (prime index {i}, {ctx.q[i].bit_length()} bits): {int((want[0] != got[r]).sum())} of {N} words differ from the oracle"
        return "ok"

    try:
        spot = parity_spot_check()
    except Exception as e:
        spot = f"ERROR: {type(e).__name__}: {e}"[:300]
    if world > 1:
        # two counters over the ranks: only a MISMATCH anywhere fails the run; a checker ERROR elsewhere is reported as such
        t = torch.tensor([1 if spot.startswith("MISMATCH") else 0, 1 if spot.startswith("ERROR") else 0], dtype=torch.int64,
                         device=_reduce_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_bad, n_err = int(t[0].item()), int(t[1].item())
        if n_bad and not spot.startswith("MISMATCH"):
            spot = f"MISMATCH on {n_bad} of {world} ranks (this rank: {spot})"
        elif n_err and spot == "ok":
            spot = f"ok on this rank; checker ERROR on {n_err} of {world} ranks (not a mismatch)"

    # Roofline of the dominant kernel, ntt_pass16_fwd_seq: the tiled pass (12 of the 16 stages) of all 30 limbs,
    # integer-class blocks first, then the fp64 class.  In the step above it follows the column pass, so it is
    # timed on its own here: lf_ntt_pass(which = 2), the library's measurement entry, launches exactly that kernel,
    # once, with the grid it has inside the full step.  (x is scratch afterwards.)
    n_roof = max(100, 2 * args.steps)      # >= 100 timed launches of the dominant kernel

    def one_pass(which):
        if ws is not None:
            check(lib.lf_ntt_pass_ws(x.data_ptr(), ws.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, 0, which,
                                     ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt_pass_ws")
            return
        check(lib.lf_ntt_pass(x.data_ptr(), B, L_LIMBS, LOGN, psi.data_ptr(), psi_dp, q_host.ctypes.data, 0, ntt_flags, which,
                              ql.data_ptr(), qh.data_ptr(), kl.data_ptr(), kh.data_ptr(), local_rank, stream), "lf_ntt_pass")

    one_pass(1)                            # (through a workspace the tiled pass reads what a column pass left there)
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
    traffic = valu = valu_busy = cols_traffic = cols_valu = None
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
        cols_valu = tj.get("ntt_fwd_cols_mixed_valu_wave_instr_per_launch")
        cols_valu = None if cols_valu is None else cols_valu * scale
    # The ceiling THIS design can reach on THIS part: every VALU instruction of a wave occupies its SIMD for 4 cycles
    # (64 lanes over 16-lane pipes; tools/ubench: fp64 FMA, 64-bit integer multiply-adds and compares all issue at that
    # rate), so a launch cannot be shorter than its wave-level VALU instructions / (1024 SIMDs x clock / 4) — at the
    # clock the package's 1 400 W cap leaves the kernel.  `issue_ceiling_frac` is that time expressed like `frac` (algorithmic
    # bytes over it, against 8 TB/s): the number 0.25-0.3 is to be read against, not the 0.60 of an HBM-bound kernel.
    issue_ceiling = None
    if valu is not None and cols_valu is not None and clocks and clocks.get("tiled_pass") and clocks.get("column_pass"):
        rate = lambda mhz: 1024 * mhz * 1e6 / 4.0       # wave-level VALU instructions per second, chip-wide
        t_tiled, t_cols = valu / rate(clocks["tiled_pass"]), cols_valu / rate(clocks["column_pass"])
        step_bytes = 16 * N * L_LIMBS * B
        issue_ceiling = {
            "tiled_pass_min_ms": 1e3 * t_tiled, "column_pass_min_ms": 1e3 * t_cols,
            "kernel_frac": alg_bytes_per_launch / t_tiled / 1e9 / HBM_PEAK_GBS,
            "whole_step_frac": step_bytes / (t_tiled + t_cols) / 1e9 / HBM_PEAK_GBS,
            "whole_step_poly_ntt_per_s": B / (t_tiled + t_cols),
            "kernel_at_fraction_of_ceiling": t_tiled / (k_ms * 1e-3),
            "whole_step_at_fraction_of_ceiling": (t_tiled + t_cols) / (dev_ms * 1e-3),
            "wave_instr_per_step": valu + cols_valu,
            "note": "min time = VALU wave-instructions (PMC, this build) / (1024 SIMDs x measured shader clock / 4 cycles)"}

    result = {
        "metric": "NTTs/sec (forward negacyclic poly-NTT, logN=16, L=30 limbs, bit-exact vs reference semantics)",
        "value": value, "unit": "poly-NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        # one polynomial of one more (untimed) step, an integer-class and an fp64-class limb, word for word vs the C oracle
        "parity_spot_check": spot,
        "config": {"workload": f"gold preset (logN=16), rows {lo}..{total - 1} of the prime chain (25 scale + base + 4 special"
                               f" primes = 30 limbs), {B} polynomials per GPU per step, forward NTT via "
                               + ("lf_ntt (C ABI, strictly in place)" if in_place else "lf_ntt_ws (C ABI: lf_ntt through a resident workspace of the operands' size)"),
                   "workspace_bytes": None if ws is None else int(ws.numel()) * 8,
                   "batch_per_gpu": B, "limbs": L_LIMBS, "logN": LOGN, "parallelism": f"replicas x{world} (independent polynomials)",
                   # repeated here because drivers keep `config` verbatim: "ok" = the timed kernel's words equal the oracle's
                   "parity_spot_check": spot},
        # What bounds the dominant kernel is VALU instruction issue, not HBM (the bytes would take half the time): `frac`
        # stays the algorithmic-bytes fraction of the 8 TB/s HBM peak the contract asks for, `issue_frac` is the measured
        # utilisation of the pipe that actually limits it (PMC; null when the counters were not taken on this build)
        "roofline": {"bound": "valu_issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "issue_frac": valu_busy, "traffic": traffic,
                     "issue_ceiling_frac": None if issue_ceiling is None else issue_ceiling["kernel_frac"],
                     "issue_ceiling": issue_ceiling,
                     "kernel": ("ntt_pass16_fwd_seq<false> (tiled pass = 12 of 16 stages" if in_place or lib.lf_tune(4, -1) != 1 else
                                "ntt_pass16_fwd_seq_ws<true> (tiled pass = 11 of 16 stages behind a 5-stage column pass")
                               + ", 16 words per thread, 8 tiles per block, all 30 limbs: 5 integer-class + 25 fp64-class)",
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
        if ws is not None:      # the same step strictly in place (lf_ntt), for the record of what the workspace buys
            for _ in range(3):
                step_in_place()
            torch.cuda.synchronize()
            extra["poly_ntt_per_s_in_place_lf_ntt"] = B / (event_time_ms(step_in_place, max(10, args.steps // 2)) * 1e-3)
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
            if ws is not None:
                check(lib.lf_ntt_ws(x.data_ptr(), ws.data_ptr(), B, L_LIMBS, LOGN, psi_c.data_ptr(), dp_c, qh_ct.ctypes.data, 0, 0,
                                    ql_c.data_ptr(), qh_c.data_ptr(), kl_c.data_ptr(), kh_c.data_ptr(), local_rank, stream), "lf_ntt_ws")
                return
            check(lib.lf_ntt(x.data_ptr(), B, L_LIMBS, LOGN, psi_c.data_ptr(), dp_c, qh_ct.ctypes.data, 0, 0, q2_c.data_ptr(),
                             ql_c.data_ptr(), qh_c.data_ptr(), kl_c.data_ptr(), kh_c.data_ptr(), local_rank, stream), "lf_ntt")
        for _ in range(5):
            step_ct()
        torch.cuda.synchronize()
        ct_ms = event_time_ms(step_ct, max(10, args.steps // 2))
        extra["poly_ntt_per_s_ciphertext_limbs_1_integer_class"] = B / (ct_ms * 1e-3)
        extra["ciphertext_limbs_note"] = (f"rows {lo_ct}..{total - 5}: 29 scale primes + base prime (a level-5 ciphertext, no special primes);"
                                          " reported for the limb-mix sensitivity only, the metric above keeps its 5 integer-class limbs")
        if ws is not None:
            # .. and its tiled pass alone: with one integer-class limb the pass is no longer bound by instruction issue but by
            # its own bytes (6 + 8 per word), which bounds what ANY cheaper integer butterfly could buy the headline
            def tiled_ct():
                check(lib.lf_ntt_pass_ws(x.data_ptr(), ws.data_ptr(), B, L_LIMBS, LOGN, psi_c.data_ptr(), dp_c, qh_ct.ctypes.data, 0, 0, 2,
                                         ql_c.data_ptr(), qh_c.data_ptr(), kl_c.data_ptr(), kh_c.data_ptr(), local_rank, stream), "lf_ntt_pass_ws")
            tiled_ct()
            torch.cuda.synchronize()
            t_ms = event_time_ms(tiled_ct, 40)
            moved = B * L_LIMBS * N * (8 + 6 * (L_LIMBS - 1) / L_LIMBS + 8 / L_LIMBS)
            extra["tiled_pass_1_integer_class"] = {"avg_launch_ms": t_ms, "moved_TB_per_s": moved / (t_ms * 1e-3) / 1e12,
                                                   "headline_tiled_pass_ms": k_ms}
        # the other halves of BASELINE's metric ("NTTs/sec at logN = 15, 16"): every limb of the silver chain at logN 15, the
        # whole gold chain incl. the inverse chain, and the reference's largest preset — each with its own roofline block
        x = ws = None      # the headline's stack and workspace are not needed any more
        torch.cuda.empty_cache()
        ntt_presets = {}
        for pname, pbatch in (("silver", 256), ("gold", 96), ("platinum", 24)):
            try:
                ntt_presets[pname] = ntt_rates_preset(pname, dev, pbatch)
            except Exception as e:
                ntt_presets[pname] = {"error": f"{type(e).__name__}: {e}"[:300]}
        result["ntt_presets"] = ntt_presets
        try:
            extra["platinum"] = platinum_rates(dev)
        except Exception as e:
            extra["platinum"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        extra["api_endpoints"] = api_endpoints(dev)
        rates, roof = engine_rates(dev, quick=False)
        extra.update(rates)
        result["roofline_engine_ops"] = roof     # cc_mult_evk / rotate_single: the metric's second half, per preset
        result["cpu_baseline"] = cpu_baseline(ctx, rows_idx, batch_cap=min(B, 64))   # bounded sample of the same workload
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
        grp = comm_block = None
        if nccl_error is not None:
            result["comm"] = {"error": "RCCL did not come up on every rank; barriers over gloo, limb-sharded legs skipped: " + nccl_error}
            result["sharded"] = "skipped: " + nccl_error
        elif not args.no_sharded:
            try:
                grp, comm_block = comm_prepare(dev, world, rank, local_rank, rccl)     # untimed channel set-up, before the watchdog
                result["comm"] = comm_block
            except Exception as e:
                result["comm"] = {"error": f"{type(e).__name__}: {e}"[:300]}

        def watchdog():
            if not done.wait(args.leg_timeout):
                extra["multi_gpu_legs_abandoned_after_s"] = args.leg_timeout
                if rank == 0:
                    print(json.dumps(result), flush=True)
                os._exit(3)     # a hung leg is a failure: the partial line is printed, the exit code says so
        threading.Thread(target=watchdog, daemon=True).start()
        multi_gpu_rates(dev, world, rank, extra, sharded=not args.no_sharded and grp is not None, grp=grp, comm_block=comm_block)
        done.set()
        # one top-level word on the limb-sharded legs (BASELINE configs[3]), so that a green replica line cannot be misread
        if "sharded" not in result:
            if args.no_sharded:
                result["sharded"] = "skipped: --no-sharded"
            elif "cc_mult_evk_gold_limb_sharded_ops_per_s" in extra:
                result["sharded"] = "ok: parity-gated against the unsharded engine on every rank, then timed (extra.*limb_sharded*)"
            else:
                result["sharded"] = "failed: " + str(extra.get("multi_gpu_limb_sharded_error") or result.get("comm", {}).get("error") or "no rate produced")
    if rccl_world1 is not None:
        result["comm"] = {"rccl_world1": rccl_world1,
                          "note": "one GPU: RCCL executed at world size 1 only (library, communicator, collectives, stream ordering, the "
                                  "engine's sharded code path on it); xGMI traffic needs N > 1"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    if spot.startswith("MISMATCH"):
        sys.exit(4)
    if world > 1 and nccl_error is not None:
        sys.exit(6)      # N > 1 without RCCL: the replica headline above is valid, the limb-sharded legs did not run — not a green run      # the headline kernel's words differ from the oracle's on this box: the line says so, the exit code too
    # (an ERROR of the checker itself — e.g. no C compiler for the oracle on the box — is reported in the line, not as a failure)


if __name__ == "__main__":
    main()
