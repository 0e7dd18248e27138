#!/bin/bash
# round-2 baseline on one box: gpu tests, bench line, kernel traces of cc_mult / rotate (silver, gold)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > $OUT/bench_base.json 2> $OUT/bench_base.err; tail -c 2500 $OUT/bench_base.json
for P in silver gold; do for OP in cc_mult rotate; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_${P}_${OP} -o kt -- python3 $REPO/tools/ccmult_profile.py $P $OP > $OUT/kt_${P}_${OP}.log 2>&1
  cd $REPO; tail -1 $OUT/kt_${P}_${OP}.log
  python3 - $OUT/kt_${P}_${OP}/kt_results.db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{r[0][:90]:90s} calls={r[1]:5d} total_us={r[2]:10.1f} avg_us={r[3]:8.1f} {r[4]:5.1f}%")
PY
done; done
