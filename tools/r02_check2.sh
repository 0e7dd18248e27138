#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
tail -c 300 gpurun_out/bench_check.json; echo
bash tools/r02_eo.sh > /dev/null 2>&1
