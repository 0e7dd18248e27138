#!/bin/bash
# One line of the box-spread table (profiles/<tag>_box_spread.txt): headline + engine-op rates of the current build on whatever box
# this call got.  Run several times as separate gpurun calls:  gpurun -- 'bash tools/lease_rates.sh'
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python3 bench.py --no-extra > gpurun_out/lease_b.json 2> gpurun_out/lease_b.err
timeout 500 python3 tools/eo.py > gpurun_out/lease_eo.log 2>&1
python3 - <<'PY'
import json, subprocess
j = json.loads([l for l in open("gpurun_out/lease_b.json").read().splitlines() if l.startswith("{")][-1])
r = json.loads([l for l in open("gpurun_out/lease_eo.log").read().splitlines() if l.startswith('{"cc_mult_evk_silver_ops')][-1])
keys = ["cc_mult_evk_silver_ops_per_s", "rotate_single_silver_ops_per_s", "cc_mult_evk_silver_batch16_ops_per_s", "cc_mult_evk_gold_ops_per_s",
        "rotate_single_gold_ops_per_s", "cc_mult_evk_gold_batch16_ops_per_s", "rotate_single_gold_batch64_rotations_per_s"]
try:
    uid = [l for l in subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True).stdout.splitlines() if "Unique ID:" in l][0].split("Unique ID:")[1].strip()
except Exception:
    uid = "?"
line = (f"LEASE uid {uid}: {j['value']:9.1f} poly-NTT/s (tiled {j['roofline']['avg_launch_ms']:.4f} ms, column {j['roofline']['column_pass_launch_ms']:.4f} ms) | "
        + " | ".join(f"{r[k]:9.1f}" for k in keys))
print(line)
open("gpurun_out/lease_line.txt", "a").write(line + "\n")
PY
