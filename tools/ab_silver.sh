#!/bin/bash
# silver / gold engine-op A/B of library variants (liberate_fhe_amd/csrc/variants/lib_<name>.so): tools/ab_silver.sh name1 name2 ..
for i in 1 2; do
  for L in "$@"; do
    for cfg in "silver cc_mult" "silver rotate" "gold cc_mult"; do
      echo "$L $(LF_HIP_LIB=$PWD/liberate_fhe_amd/csrc/variants/lib_$L.so python tools/ccmult_profile.py $cfg 2>/dev/null | tail -1)"
    done
  done
done
