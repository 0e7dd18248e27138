#!/usr/bin/env python3
"""gpurun_out/ of tools/pmc_round.sh <tag> -> the tracked summaries under profiles/ (python3 tools/summarize_round.py <tag>):

  <tag>_bench_kernel_stats.txt   rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extra`
  <tag>_bench_pmc.txt          FETCH_SIZE / WRITE_SIZE / SQ passes over the same command (separate runs)
  traffic_<tag>.json           what bench.py folds into its roofline block (bytes and VALU figures per launch)
  <tag>_engine_ops_pmc.txt       the same counters per kernel of gold / silver cc_mult (+relinearize)

Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts a 16 B/lane coalesced read at
half its bytes (MI355X_MICROARCH.md, HBM section) -> doubled.  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count in units of
4 cycles summed over the chip's 1024 SIMDs; SQ_BUSY_CU_CYCLES in cycles summed over the 256 CUs."""
import collections, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
sys.path.insert(0, ROOT)
import __graft_entry__ as _g   # noqa: E402
DIGEST = _g.library_digest()   # the sources these counters were taken at: bench.py drops the figures when the build differs


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(tag):
    """{(kernel, grid): {counter: avg, 'us': avg, 'n': dispatches}} of one PMC pass"""
    db = os.path.join(src, f"pmc2_{tag}", "p_results.db")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, n, v, du, gx in q(db, "select kernel_name, counter_name, value, duration, grid_size_x from counters_collection"):
        agg[(short(k), gx)][n].append(v)
        agg[(short(k), gx)]["us"].append(du / 1e3)
    return {k: {**{n: sum(v) / len(v) for n, v in d.items()}, "n": len(d["us"])} for k, d in agg.items()}


# ---- bench: kernel stats -------------------------------------------------------------------------------------------
lines = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra   ({TAG})",
         "# columns: kernel | calls | total_us | avg_us | % of GPU time", ""]
for name, calls, total, avg, pct in q(os.path.join(src, f"prof_{TAG}", "stats_results.db"),
                                      "select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append(f"{short(name)[:90]:90s} | {calls:6d} | {total:12.1f} | {avg:10.2f} | {pct:6.2f}")
try:
    bench = json.load(open(os.path.join(src, f"bench_{TAG}.json")))
    r = bench["roofline"]
    lines += ["", f"# bench.py (un-profiled run of the same box): value {bench['value']:.0f} poly-NTT/s, ms_per_step {bench['ms_per_step']:.4f};",
              f"# roofline leg (lf_ntt_pass which = 2, {r['launches_timed']} launches): avg_launch_ms {r['avg_launch_ms']:.4f} from HIP events;",
              f"# the last {r['launches_timed']} tiled-pass dispatches (ntt_pass16_fwd_seq_ws) of the trace above are that leg."]
    durs = [d[0] / 1e3 for d in q(os.path.join(src, f"prof_{TAG}", "stats_results.db"),
                                  "select duration from kernels where (name like '%ntt_fwd_pass_mixed<false>%' or name like '%ntt_pass16_mixed<false, false>%' or name like '%ntt_pass16_fwd_seq<false>%' or name like '%ntt_pass16_fwd_seq_ws%') order by start")]
    tail = durs[-int(r["launches_timed"]):]
    lines.append(f"# their average under rocprofv3: {sum(tail) / len(tail):.2f} us (min {min(tail):.2f}, max {max(tail):.2f})")
except Exception as e:
    lines.append(f"# bench line not found: {e}")
open(os.path.join(out, f"{TAG}_bench_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))

# ---- bench: PMC ----------------------------------------------------------------------------------------------------
f, w, v = counters("bench_fetch"), counters("bench_write"), counters("bench_valu")
pm = [f"# rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --no-extra --steps 5 --warmup 2   ({TAG}); one run per line group",
      "# kernel/grid | us | FETCH_SIZE KiB (raw) | read MB (x2, gfx950) | WRITE_SIZE KiB | write MB | moved TB/s | VALU wave-instr | VALU busy | CU busy", ""]
traffic = {"batch_per_gpu": 128, "library_digest": DIGEST}
for key in sorted(f):
    if "ntt_" not in key[0] or "cols_mixed" not in key[0] and "cols_ws" not in key[0] and "pass16" not in key[0] and "fwd_pass" not in key[0]:
        continue
    rd, wr = 2 * f[key]["FETCH_SIZE"] * 1024, w[key]["WRITE_SIZE"] * 1024
    us = v[key]["us"]
    clk_cu = v[key]["SQ_BUSY_CU_CYCLES"] / 256 / us / 1e3     # GHz while busy (upper bound: busy CUs only)
    valu_busy = v[key]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v[key]["SQ_BUSY_CU_CYCLES"] / 256)
    pm.append(f"{key[0]}/g{key[1]} | {us:8.1f} | {f[key]['FETCH_SIZE']:10.1f} | {rd / 1e6:8.1f} | {w[key]['WRITE_SIZE']:10.1f} | {wr / 1e6:8.1f} | "
              f"{(rd + wr) / us / 1e6:5.2f} | {v[key]['SQ_INSTS_VALU']:.4g} | {valu_busy:5.3f} | cycles/CU {v[key]['SQ_BUSY_CU_CYCLES'] / 256:.4g} (~{clk_cu:.2f} GHz x launch)")
    name = "ntt_fwd_pass_mixed" if ("fwd_pass" in key[0] or "pass16" in key[0]) else "ntt_fwd_cols_mixed"
    traffic[f"{name}_bytes_per_launch"] = rd + wr
    traffic[f"{name}_valu_wave_instr_per_launch"] = v[key]["SQ_INSTS_VALU"]
    traffic[f"{name}_valu_busy_frac"] = valu_busy
    traffic[f"{name}_salu_per_valu"] = v[key]["SQ_INSTS_SALU"] / v[key]["SQ_INSTS_VALU"]
traffic["note"] = ("per launch at batch_per_gpu polynomials x 30 limbs; bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes, gfx950 half-count "
                   "correction on reads); valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over SQ_BUSY_CU_CYCLES / 256 CUs; source "
                   f"profiles/{TAG}_bench_pmc.txt")
open(os.path.join(out, f"{TAG}_bench_pmc.txt"), "w").write("\n".join(pm) + "\n")
json.dump(traffic, open(os.path.join(out, f"traffic_{TAG}.json"), "w"), indent=1)
print("\n".join(pm))

# ---- engine ops: PMC per kernel ------------------------------------------------------------------------------------
eo = [f"# rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/ccmult_profile.py <preset> <op> --mark   ({TAG}); FETCH / WRITE / SQ in separate runs",
      "# averages over all dispatches of the run (40 warm-up + 10 timed ops); a kernel listed twice runs on two grids per op",
      "# kernel/grid | us | read MB (2 x FETCH_SIZE) | write MB | moved TB/s | VALU wave-instr | VALU busy | launches per op"]
eo_json = {"library_digest": DIGEST,
           "note": ("per kernel and grid: average launch (us), HBM bytes (reads = 2 x FETCH_SIZE KiB, the gfx950 correction; writes = WRITE_SIZE KiB), "
                    "moved TB/s, VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over SQ_BUSY_CU_CYCLES / 256 CUs; bytes_per_op = sum over the "
                    f"op's launches; source profiles/{TAG}_engine_ops_pmc.txt")}
for preset in ("gold", "silver"):
    for op, tag in (("cc_mult", preset), ("rotate", preset + "_rot")):
        try:
            f, w, v = counters(f"{tag}_fetch"), counters(f"{tag}_write"), counters(f"{tag}_valu")
        except Exception as e:
            eo.append(f"## {preset} {op}: missing ({e})")
            continue
        eo += ["", f"## {preset} {op}" + (" + relinearize" if op == "cc_mult" else "_single")]
        ops = max((d["n"] for k, d in v.items() if k[0].startswith("ks_inner2_kernel")), default=0)   # one inner product per op
        kernels, total_bytes = [], 0.0
        for key in sorted(v, key=lambda k: -v[k]["us"] * v[k]["n"]):
            if not (key[0].startswith(("ntt_", "ks_", "tensor", "rescale"))) or key not in f or key not in w:
                continue
            rd, wr, us = 2 * f[key]["FETCH_SIZE"] * 1024, w[key]["WRITE_SIZE"] * 1024, v[key]["us"]
            busy = v[key]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v[key]["SQ_BUSY_CU_CYCLES"] / 256)
            per_op = round(v[key]["n"] / ops) if ops else 1
            if per_op < 1:
                continue            # set-up launches (key conversion, table builds): not part of an op
            eo.append(f"{key[0]}/g{key[1]} | {us:7.1f} | {rd / 1e6:7.1f} | {wr / 1e6:7.1f} | {(rd + wr) / us / 1e6:5.2f} | {v[key]['SQ_INSTS_VALU']:.4g} | {busy:5.3f} | {per_op}")
            kernels.append({"kernel": key[0], "grid": key[1], "us": us, "per_op": per_op, "read_MB": rd / 1e6, "write_MB": wr / 1e6,
                            "moved_TBps": (rd + wr) / us / 1e6, "valu_busy": busy, "valu_wave_instr": v[key]["SQ_INSTS_VALU"]})
            total_bytes += (rd + wr) * per_op
        eo.append(f"# HBM bytes per op (sum over its launches): {total_bytes / 1e6:.1f} MB")
        eo_json[f"{preset}_{op}"] = {"kernels": kernels, "bytes_per_op": total_bytes}
# batched groups (gold, 16 ciphertexts under one key = 4 launch sets of nct = 4): bytes per CIPHERTEXT
for op, tag in (("rotate_batch", "gold_rotate_batch"), ("cc_mult_batch", "gold_cc_mult_batch")):
    try:
        f, w, v = counters(f"{tag}_fetch"), counters(f"{tag}_write"), counters(f"{tag}_valu")
    except Exception as e:
        eo.append(f"## gold {op}: missing ({e})")
        continue
    eo += ["", f"## gold {op}: 16 ciphertexts under one key, groups of 4 per key-switch launch set (launches per GROUP in the last column)"]
    groups = max((d["n"] for k, d in v.items() if k[0].startswith("ks_inner2_kernel")), default=0)   # one inner product per group
    kernels, total_bytes = [], 0.0
    for key in sorted(v, key=lambda k: -v[k]["us"] * v[k]["n"]):
        if not (key[0].startswith(("ntt_", "ks_", "tensor", "rescale"))) or key not in f or key not in w:
            continue
        rd, wr, us = 2 * f[key]["FETCH_SIZE"] * 1024, w[key]["WRITE_SIZE"] * 1024, v[key]["us"]
        busy = v[key]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v[key]["SQ_BUSY_CU_CYCLES"] / 256)
        per_grp = v[key]["n"] / groups if groups else 1
        if per_grp < 0.5:
            continue
        eo.append(f"{key[0]}/g{key[1]} | {us:7.1f} | {rd / 1e6:7.1f} | {wr / 1e6:7.1f} | {(rd + wr) / us / 1e6:5.2f} | {v[key]['SQ_INSTS_VALU']:.4g} | {busy:5.3f} | {per_grp:.2f}")
        kernels.append({"kernel": key[0], "grid": key[1], "us": us, "per_op": per_grp / 4, "read_MB": rd / 1e6, "write_MB": wr / 1e6,
                        "moved_TBps": (rd + wr) / us / 1e6, "valu_busy": busy, "valu_wave_instr": v[key]["SQ_INSTS_VALU"]})
        total_bytes += (rd + wr) * per_grp / 4
    eo.append(f"# HBM bytes per ciphertext (sum over a group's launches / 4): {total_bytes / 1e6:.1f} MB")
    eo_json[f"gold_{op}"] = {"kernels": kernels, "bytes_per_op": total_bytes}
open(os.path.join(out, f"{TAG}_engine_ops_pmc.txt"), "w").write("\n".join(eo) + "\n")
json.dump(eo_json, open(os.path.join(out, f"{TAG}_engine_ops_pmc.json"), "w"), indent=1)
print("\n".join(eo))
