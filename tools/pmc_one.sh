#!/bin/bash
# one PMC pass over the bench kernels: tools/pmc_one.sh <tag> <counters...>   (run on the GPU box through gpurun)
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
ARGS="${BENCH_ARGS:---no-extra --steps 3 --warmup 1}"
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$TAG -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$TAG.log 2>&1
cd $REPO
python3 tools/pmc_report.py "gpurun_out/pmc_$TAG/*_results.db"
