#!/bin/bash
# HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the headline step, two-launch form vs one-launch team form
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
for MODE in ${MODES:-0 1}; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmcT_${MODE}_$C
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d $OUT/pmcT_${MODE}_$C -o p -- python3 $REPO/tools/team_step.py ${MODE%%_*} ${GRIDB:-1024} > $OUT/pmcT_${MODE}_$C.log 2>&1)
done; done
python3 - <<'PY'
import sqlite3, collections, glob, os
out = os.path.join(os.getcwd(), "gpurun_out")
for mode in os.environ.get('MODES', '0 1').split():
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        db = glob.glob(os.path.join(out, f"pmcT_{mode}_{c}", "**", "p_results.db"), recursive=True)
        if not db:
            print(mode, c, "no db"); continue
        con = sqlite3.connect(db[0])
        agg = collections.defaultdict(list)
        for k, n, v, du in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            if "ntt" in k:
                agg[k.split("(")[0][-60:]].append((v, du))
        for k, vs in agg.items():
            vs = vs[len(vs) // 2:]
            kib = sum(v for v, _ in vs) / len(vs)
            us = sum(d for _, d in vs) / len(vs) / 1e3
            mult = 2 if c == "FETCH_SIZE" else 1    # gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes
            print(f"mode {mode} {c:10s} {k:60s} {kib * 1024 * mult / 1e9:7.3f} GB per launch ({'x2 corrected' if mult == 2 else 'as counted'}), {us:8.1f} us")
PY
rm -rf $OUT/pmcT_*
