#!/bin/bash
# headline-step A/B of library variants: tools/ab_ntt.sh name1 name2 ..   ("base" = the built library)
for i in 1 2; do
  for L in "$@"; do
    if [ "$L" = base ]; then LIB=""; else LIB=$PWD/liberate_fhe_amd/csrc/variants/lib_$L.so; fi
    LF_HIP_LIB=$LIB python bench.py --no-extra --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$L', 'value %.0f' % j['value'], 'step_ms %.4f' % j['ms_per_step'], 'tile_ms %.4f' % r['avg_launch_ms'], 'cols_ms %.4f' % r['column_pass_launch_ms'])"
  done
done
