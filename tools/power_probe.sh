#!/bin/bash
# package power / clocks reported by the SMI while the tiled pass (or the column pass) runs in a loop (development aid)
#   tools/power_probe.sh            -> gpurun_out/smi_samples.txt
mkdir -p gpurun_out
python - <<'PY' &
import os, sys, time, warnings
sys.path.insert(0, os.getcwd()); warnings.filterwarnings("ignore")
sys.argv = ["clock_probe"]
import runpy
ns = runpy.run_path("tools/clock_probe.py")          # builds the buffers, prints its own clock table first
import torch
one_pass, full = ns["one_pass"], ns["full"]
for name, fn, n in (("tiled", lambda: one_pass(2), 6000), ("cols", lambda: one_pass(1), 9000), ("full", full, 3500)):
    print("LOOP", name, time.time(), flush=True)
    for i in range(n):
        fn()
        if i % 200 == 199: torch.cuda.synchronize()
    torch.cuda.synchronize()
print("LOOP end", time.time(), flush=True)
PY
PY_PID=$!
for i in $(seq 1 90); do
  echo "T $(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Package Power\|sclk' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
  kill -0 $PY_PID 2>/dev/null || break
done > gpurun_out/smi_samples.txt
wait $PY_PID
tail -60 gpurun_out/smi_samples.txt
