#!/bin/bash
# round 5, first GPU call: test-suite on the round's first changes + the co-run probe
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python tools/corun_probe.py > gpurun_out/corun_probe.txt 2>&1
echo "corun rc=$?"; cat gpurun_out/corun_probe.txt | tail -60
