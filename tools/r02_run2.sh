#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25
timeout 600 python bench.py > $OUT/bench_r02a.json 2> $OUT/bench_r02a.err; tail -c 6000 $OUT/bench_r02a.json; tail -3 $OUT/bench_r02a.err
bash tools/profile_engine_ops.sh
