#!/bin/bash
# engine-op A/B of library variants: tools/ab_engine_ops.sh lib1 lib2 ...  (gold + silver, cc_mult + rotate)
set -u
for i in 1 2; do
  for L in "$@"; do
    for P in gold silver; do for OP in rotate cc_mult; do
      echo "$L $(LF_HIP_LIB=$PWD/$L python tools/ccmult_profile.py $P $OP 2>/dev/null | tail -1)"
    done; done
  done
done
