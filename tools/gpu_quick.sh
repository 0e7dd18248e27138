#!/bin/bash
# GPU tests (stop at first failure) + headline step + engine-op rates
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
bash tools/ab_ntt.sh base
timeout 600 python tools/eo.py > gpurun_out/eo.log 2>&1; tail -2 gpurun_out/eo.log | head -1
