#!/usr/bin/env python3
"""rocprofv3 kernel traces of the engine ops (tools/profile_engine_ops.sh) -> profiles/<tag>_engine_ops_kernel_stats.txt
and profiles/<tag>_engine_ops_summary.json (read by bench.py: dominant kernel and per-kernel time per op).

Only the STEADY-STATE window counts: the dispatches between the two marker launches that tools/ccmult_profile.py --mark
puts around its timed loop (ew_kernel with a 1-block grid).  Everything before it is set-up (synthetic inputs,
tables, warm-up).  The summary also records how many __amd_rocclr_copyBuffer dispatches fall inside the window."""
import json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
sys.path.insert(0, ROOT)
import __graft_entry__ as _g   # noqa: E402
src = os.path.join(ROOT, "gpurun_out")
N_OPS = 10
lines, summary = [], {}
for preset in ("silver", "gold"):
    for op in ("cc_mult", "rotate"):
        db = os.path.join(src, f"eo_{preset}_{op}", "kt_results.db")
        if not os.path.exists(db):
            continue
        con = sqlite3.connect(db)
        rows = con.execute("select name, start, end, grid_x from kernels order by start").fetchall()
        marks = [i for i, r in enumerate(rows) if "ew_kernel" in r[0] and r[3] <= 256]
        if len(marks) < 2:
            lines.append(f"## {preset} {op}: markers not found ({len(marks)})")
            continue
        win = rows[marks[-2] + 1:marks[-1]]
        agg = {}
        for name, st, en, gx in win:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = agg.setdefault(short, [0, 0.0])
            a[0] += 1
            a[1] += (en - st) / 1e3
        total = sum(v[1] for v in agg.values())
        span = (win[-1][2] - win[0][1]) / 1e3
        lines += [f"## {preset} {op}: steady-state window = {N_OPS} ops, {len(win)} dispatches, kernel time {total / N_OPS:.1f} us/op, "
                  f"span {span / N_OPS:.1f} us/op (rocprofv3 --kernel-trace, tools/profile_engine_ops.sh)",
                  "# kernel | dispatches per op | us per op | % of kernel time"]
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{k[:70]:70s} | {c / N_OPS:5.1f} | {t / N_OPS:8.1f} | {100 * t / total:5.1f}")
        copies = sum(v[0] for k, v in agg.items() if "copyBuffer" in k)
        lines += [f"# __amd_rocclr_copyBuffer dispatches inside the window: {copies}", ""]
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        summary[f"{preset}_{op}"] = {
            "source": f"profiles/{tag}_engine_ops_kernel_stats.txt", "kernel_us_per_op": total / N_OPS, "span_us_per_op": span / N_OPS,
            "dispatches_per_op": len(win) / N_OPS, "copy_dispatches_in_steady_state": copies,
            "dominant_kernel": dom[0], "dominant_kernel_us_per_op": dom[1][1] / N_OPS,
            "dominant_kernel_avg_us": dom[1][1] / dom[1][0],
            "kernels_us_per_op": {k: round(v[1] / N_OPS, 2) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
open(os.path.join(ROOT, "profiles", f"{tag}_engine_ops_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
summary["library_digest"] = _g.library_digest()   # bench.py drops these figures when the running build differs
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_engine_ops_summary.json"), "w"), indent=1)
print("\n".join(lines))
