#!/usr/bin/env python3
"""Turn the rocprofv3 result databases written by tools/profile_bench.sh into the small text / JSON
summaries kept under profiles/ (the .db files themselves stay in gpurun_out/, which is scratch).

    python tools/summarize_profile.py r01
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "profiles")
src = os.path.join(ROOT, "gpurun_out")


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


lines = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra   ({tag})",
         "# columns: kernel | calls | total_us | avg_us | % of GPU time", ""]
for name, calls, total, avg, pct in q(os.path.join(src, f"prof_{tag}", "stats_results.db"),
                                      "select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append(f"{name[:110]:110s} | {calls:6d} | {total:12.1f} | {avg:10.2f} | {pct:6.2f}")
# the roofline leg of bench.py times the dominant kernel on its own (last `launches_timed` dispatches of it);
# inside the full step the same kernel overlaps the integer-class launches of the side stream and runs longer
try:
    bench = json.load(open(os.path.join(src, f"bench_{tag}.json")))
    n_last = int(bench["roofline"].get("launches_timed", 0))
except Exception:
    n_last = 0
if n_last:
    rows = q(os.path.join(src, f"prof_{tag}", "stats_results.db"),
             "select duration from kernels where name like '%ntt_fwd_pass_mixed%' order by dispatch_id")
    durs = [r[0] / 1e3 for r in rows]
    if len(durs) >= n_last:
        tail = durs[-n_last:]
        # bench.py's roofline leg: 3 warm-up + n_last timed launches of the tiled pass alone, then 1 + n_last
        # launches of the column pass (a different kernel): the tiled-pass kernel's last n_last dispatches are the timed ones
        lines += ["", f"# ntt_fwd_pass_mixed, roofline leg only (last {n_last} dispatches: the kernel alone on the GPU, "
                      f"lf_ntt_pass which=2): avg {sum(tail) / len(tail):.2f} us, min {min(tail):.2f}, max {max(tail):.2f}",
                  f"# bench.py reported avg_launch_ms = {bench['roofline']['avg_launch_ms'] * 1e3:.2f} us from HIP events (un-profiled run)"]
open(os.path.join(out_dir, f"{tag}_bench_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# HBM traffic from the PMC passes.  Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a 16 B/lane coalesced read at
# half its bytes -> doubled here.  Calibrated in the same run on ew_kernel<1> (in-place mont_enter of a
# 39 x 65536 x 8 B tensor = 20,447,232 B): FETCH_SIZE*1024*2 and WRITE_SIZE*1024 both reproduce it.
pm = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-extra --steps 5 --warmup 2   ({tag})",
      "# per-dispatch values for the NTT pass kernel, in KiB as reported; bytes = KiB*1024 (x2 for FETCH_SIZE on gfx950)", ""]
per = {}
for counter, sub, stem in (("FETCH_SIZE", f"pmc_fetch_{tag}", "fetch"), ("WRITE_SIZE", f"pmc_write_{tag}", "write")):
    db = os.path.join(src, sub, f"{stem}_results.db")
    rows = q(db, "select kernel_name, grid_size_x, count(*), avg(value), min(value), max(value), avg(duration) "
                 "from counters_collection group by kernel_name")
    for r in rows:
        pm.append(f"{counter:10s} | {r[0][:70]:70s} | n={r[2]:4d} | avg={r[3]:12.1f} | min={r[4]:12.1f} | max={r[5]:12.1f} | avg_ns={r[6]:10.0f}")
    disp = q(db, "select dispatch_id, value, duration from counters_collection where kernel_name like '%ntt_fwd_pass_mixed%' order by dispatch_id")
    per[counter] = disp
    per[counter + "_cols"] = q(db, "select dispatch_id, value, duration from counters_collection where kernel_name like '%ntt_fwd_cols_mixed%' order by dispatch_id")
    pm.append("")
pm.append("# ntt_fwd_pass_mixed (30 limbs x the bench batch): the tiled pass (12 stages + twiddles), one dispatch per transform")
for (d, v, ns), (_, w, _) in zip(per["FETCH_SIZE"], per["WRITE_SIZE"]):
    pm.append(f"dispatch {d}: FETCH_SIZE={v:10.1f} KiB -> {2 * v * 1024 / 1e6:7.1f} MB read (corrected) | WRITE_SIZE={w:10.1f} KiB -> {w * 1024 / 1e6:7.1f} MB | {ns / 1e3:7.1f} us")
pm.append("# ntt_fwd_cols_mixed<4> (same limbs): the column pass (4 leading stages), one dispatch per transform")
for (d, v, ns), (_, w, _) in zip(per["FETCH_SIZE_cols"], per["WRITE_SIZE_cols"]):
    pm.append(f"dispatch {d}: FETCH_SIZE={v:10.1f} KiB -> {2 * v * 1024 / 1e6:7.1f} MB read (corrected) | WRITE_SIZE={w:10.1f} KiB -> {w * 1024 / 1e6:7.1f} MB | {ns / 1e3:7.1f} us")
open(os.path.join(out_dir, f"{tag}_bench_pmc_hbm.txt"), "w").write("\n".join(pm) + "\n")

fetch = sum(v for _, v, _ in per["FETCH_SIZE"]) / len(per["FETCH_SIZE"])
write = sum(v for _, v, _ in per["WRITE_SIZE"]) / len(per["WRITE_SIZE"])
try:
    batch = int(json.load(open(os.path.join(src, f"bench_{tag}.json")))["config"]["batch_per_gpu"])
except Exception:
    batch = 128
traffic = {"ntt_fwd_pass_mixed_bytes_per_launch": (2 * fetch + write) * 1024, "batch_per_gpu": batch,
           "fetch_kib_avg_raw": fetch, "write_kib_avg": write,
           "note": "ntt_fwd_pass_mixed (tiled pass, 30 limbs x `batch_per_gpu` polynomials) per launch; FETCH_SIZE doubled per the gfx950 1/2-count caveat"}
cf, cw = per["FETCH_SIZE_cols"], per["WRITE_SIZE_cols"]
if cf and cw:
    traffic["ntt_fwd_cols_mixed_bytes_per_launch"] = (2 * sum(v for _, v, _ in cf) / len(cf) + sum(v for _, v, _ in cw) / len(cw)) * 1024
# keep the instruction-count entries of an earlier PMC pass (tools/pmc_one.sh) if this run did not produce them
try:
    prev = json.load(open(os.path.join(out_dir, f"traffic_{tag}.json")))
    for k in ("ntt_fwd_pass_mixed_valu_wave_instr_per_launch", "valu_note"):
        if k in prev and k not in traffic:
            traffic[k] = prev[k]
except Exception:
    pass
json.dump(traffic, open(os.path.join(out_dir, f"traffic_{tag}.json"), "w"), indent=1)
print(traffic)
