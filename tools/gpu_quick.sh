#!/bin/bash
# GPU tests (stop at first failure) + engine-op rates + host enqueue time
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python tools/eo.py > gpurun_out/eo.log 2>&1; tail -2 gpurun_out/eo.log
timeout 600 python tools/host_overhead.py 2>&1 | tee gpurun_out/host_overhead.log | tail -4
