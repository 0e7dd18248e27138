"""Per-kernel averages of the PMC passes written by tools/pmc_sq.sh (development aid)."""
import sqlite3, glob, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_sq*/*_results.db"
for db in sorted(glob.glob(pat)):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, value, duration, grid_size_x from counters_collection").fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, n, v, du, gx in rows:
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48] + f"/g{gx}"
        agg[short][n].append(v)
        agg[short]["us"].append(du / 1e3)
    for k in sorted(agg):
        if "ntt_" not in k and "ks_" not in k:
            continue
        print(db.split("/")[-2], k, {n: round(sum(v) / len(v), 1) for n, v in agg[k].items()}, "n=%d" % len(agg[k]["us"]))
