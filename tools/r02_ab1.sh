#!/bin/bash
# A/B round: parity of the working tree, instruction rates, then bench + engine ops per library variant
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
./tools/ubench/ubench > $OUT/ubench_r02.txt 2>&1; cat $OUT/ubench_r02.txt
V=liberate_fhe_amd/csrc/variants
for i in 1 2; do
  for L in $V/lib_base.so $V/lib_x1.so liberate_fhe_amd/csrc/libckks_hip.so; do
    LF_HIP_LIB=$PWD/$L python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value']), 'tiled_us', round(r['avg_launch_ms']*1e3,1), 'cols_us', round(r['column_pass_launch_ms']*1e3,1))"
  done
done
for L in $V/lib_base.so $V/lib_x1.so liberate_fhe_amd/csrc/libckks_hip.so; do
  for P in silver gold; do for OP in cc_mult rotate; do
    echo -n "$L "; LF_HIP_LIB=$PWD/$L python tools/ccmult_profile.py $P $OP 2>/dev/null | tail -1
  done; done
done
