#!/bin/bash
# kernel times of gold rotate for library variants: tools/kt_gold_rotate.sh name1 name2 ..
for L in "$@"; do
  echo "== $L"
  LF_HIP_LIB=$PWD/liberate_fhe_amd/csrc/variants/lib_$L.so bash tools/ktrace_cmd.sh v_$L python3 $PWD/tools/ccmult_profile.py gold rotate 2>&1 | grep -E "pass16|ks_ext|inner2" | head -4
done
