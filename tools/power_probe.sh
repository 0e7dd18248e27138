#!/bin/bash
# power / clocks reported by the SMI while the tiled pass runs in a loop (development aid)
rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
( for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/smi_samples.txt 2>&1 &
SMI=$!
python tools/clock_probe.py 2>&1 | tail -5
wait $SMI
cat gpurun_out/smi_samples.txt | cut -c1-400
