#!/bin/bash
# A/B of two builds of libckks_hip.so in ONE GPU-box call (boxes differ by a few percent):
#   tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do
  for L in "$A" "$B"; do
    LF_HIP_LIB=$PWD/$L python bench.py --no-extra "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value']), round(r['avg_launch_ms']*1e3,1), round(r['column_pass_launch_ms']*1e3,1))"
  done
done
