#!/bin/bash
# one PMC pass over an engine op: tools/pmc_op.sh <tag> <preset> <op> <counters...>   (run on the GPU box through gpurun)
set -u
export TMPDIR=/tmp
TAG=$1; PRESET=$2; OP=$3; shift 3
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$TAG -o $TAG -- python3 $REPO/tools/ccmult_profile.py $PRESET $OP > $OUT/pmc_$TAG.log 2>&1
cd $REPO
python3 tools/pmc_report.py "gpurun_out/pmc_$TAG/*_results.db"
