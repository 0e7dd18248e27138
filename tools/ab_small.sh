#!/bin/bash
# bronze / silver engine-op A/B of library variants: tools/ab_small.sh name1 name2 ..
for i in 1 2; do
  for L in "$@"; do
    for cfg in "bronze cc_mult" "bronze rotate" "silver cc_mult" "silver rotate"; do
      echo "$L $(LF_HIP_LIB=$PWD/liberate_fhe_amd/csrc/variants/lib_$L.so python tools/ccmult_profile.py $cfg 2>/dev/null | tail -1)"
    done
  done
done
