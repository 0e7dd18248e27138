#!/bin/bash
# per-kernel time of an arbitrary python command under rocprofv3: tools/ktrace_cmd.sh <tag> python3 script.py args..
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_$TAG -o kt -- "$@" > $OUT/kt_$TAG.log 2>&1
cd $REPO && python3 - <<PY
import sqlite3
con = sqlite3.connect('gpurun_out/kt_$TAG/kt_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{r[0].replace('(anonymous namespace)::','')[:70]:70s} calls={r[1]:5d} total_us={r[2]/1e3:10.1f} avg_us={r[3]/1e3:8.1f} {r[4]:5.1f}%")
PY
