#!/bin/bash
# bench-only A/B of several library variants: tools/ab_bench.sh lib1 lib2 ...
set -u
for i in 1 2; do
  for L in "$@"; do
    LF_HIP_LIB=$PWD/$L python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value']), 'step_ms', round(d['ms_per_step'],3), 'tiled_us', round(r['avg_launch_ms']*1e3,1), 'cols_us', round(r['column_pass_launch_ms']*1e3,1))"
  done
done
