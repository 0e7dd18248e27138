#!/bin/bash
# per-kernel average durations of one bench configuration (run on the GPU box): tools/kstat.sh <tag> [bench args]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kstat_$TAG -o k -- python3 $REPO/bench.py --no-extra "$@" > $OUT/kstat_$TAG.log 2>&1
cd $REPO
python3 - "$OUT/kstat_$TAG/k_results.db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in con.execute("select * from top_kernels limit 8"):
    print(f"{name[:70]:70s} calls {calls:5d} avg {avg:9.1f} us  {pct:5.1f}%")
PY
