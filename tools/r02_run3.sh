#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_distributed_gpu.py tests/test_keygen_golden.py -m gpu -x -q 2>&1 | tail -5
LF_BENCH_REHEARSE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29911 bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 > $OUT/bench_rehearse.json 2> $OUT/bench_rehearse.err
tail -c 1800 $OUT/bench_rehearse.json; echo; tail -5 $OUT/bench_rehearse.err
