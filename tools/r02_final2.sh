#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 bench.py > $OUT/bench_r02.json 2> $OUT/bench_r02.err; tail -c 300 $OUT/bench_r02.json; echo
bash tools/profile_engine_ops.sh
LF_BENCH_REHEARSE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29911 bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 > $OUT/bench_rehearse.json 2> $OUT/bench_rehearse.err
python3 -c "
import json; d=json.load(open('$OUT/bench_rehearse.json')); print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d['extra'].items()})"
