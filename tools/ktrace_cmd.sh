#!/bin/bash
# per-kernel time of any python tool under rocprofv3: tools/ktrace_cmd.sh tools/rotate_batch_bench.py gold 16
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/ktc -o ktc -- python3 $REPO/"$@" > $OUT/ktc.log 2>&1
tail -2 $OUT/ktc.log
cd $REPO && python3 - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/ktc/ktc_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
    print(f"{r[0][:80]:80s} calls={r[1]:5d} total_us={r[2]:10.1f} avg_us={r[3]:8.1f} {r[4]:5.1f}%")
PY
