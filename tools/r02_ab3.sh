#!/bin/bash
set -u
V=liberate_fhe_amd/csrc/variants
A=${1:-$V/lib_base2.so}; B=${2:-liberate_fhe_amd/csrc/libckks_hip.so}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
  for L in $A $B; do
    LF_HIP_LIB=$PWD/$L python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value']), 'step_ms', round(d['ms_per_step'],3), 'tiled_us', round(r['avg_launch_ms']*1e3,1), 'cols_us', round(r['column_pass_launch_ms']*1e3,1))"
  done
done
for i in 1 2; do for L in $A $B; do
  for P in silver gold; do for OP in cc_mult rotate; do
    echo -n "$L "; LF_HIP_LIB=$PWD/$L python tools/ccmult_profile.py $P $OP 2>/dev/null | tail -1
  done; done
done; done
