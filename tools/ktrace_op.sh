#!/bin/bash
# per-kernel time of one engine op under rocprofv3: ./tools/ktrace_op.sh gold cc_mult
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
python3 $REPO/tools/ccmult_profile.py $1 $2
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kto -o kto -- python3 $REPO/tools/ccmult_profile.py $1 $2 > $OUT/kto.log 2>&1
cd $REPO && python3 - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/kto/kto_results.db')
tot = 0
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{r[0][:80]:80s} calls={r[1]:5d} total_us={r[2]:10.1f} avg_us={r[3]:8.1f} {r[4]:5.1f}%")
PY
