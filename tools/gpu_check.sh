#!/bin/bash
# One GPU-box call: GPU test-suite, the 2-rank rehearsal of `bench.py --gpus 2` (self-spawned ranks, parity gate,
# host-staged point-to-point transport), then the N = 1 bench line.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
LF_BENCH_REHEARSE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_rehearse2.json 2> gpurun_out/bench_rehearse2.err
echo "rehearsal rc=$?"
tail -c 1500 gpurun_out/bench_rehearse2.json
tail -5 gpurun_out/bench_rehearse2.err
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1])
print("value", j["value"], "roofline", j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["roofline"]["column_pass_launch_ms"])
print({k: round(v, 1) for k, v in j["extra"].items() if isinstance(v, float)})
PY
