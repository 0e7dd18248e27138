#!/bin/bash
# full GPU suite + default bench on the current build
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
tail -c 2500 gpurun_out/bench_check.json; tail -3 gpurun_out/bench_check.err
