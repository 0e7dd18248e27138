#!/bin/bash
# A/B of two library builds on the engine ops (run on the GPU box): tools/ab_op.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    for cfg in "silver cc_mult" "gold cc_mult" "gold rotate"; do
      echo -n "$L: "; LF_HIP_LIB=$PWD/$L python tools/ccmult_profile.py $cfg 2>&1 | tail -1
    done
  done
done
