#!/bin/bash
# per-kernel durations of the bench step (rocprofv3 kernel trace), printed as a table
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $REPO/bench.py --no-extra --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $REPO && python3 - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/kt/kt_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"): print(r[0][:90], r[1:])
rows = con.execute("select * from kernels limit 1")
cols = [d[0] for d in rows.description]; print(cols)
PY
python3 - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/kt/kt_results.db')
rows = con.execute("select name, grid_x, duration, dispatch_id from kernels where name like '%ntt_%pass%' order by dispatch_id").fetchall()
for r in rows[-8:]: print(('dp ' if '<true>' in r[0] else 'int'), r[1], round(r[2]/1e3,1), 'us')
PY
