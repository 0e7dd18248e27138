#!/bin/bash
# Build a variant of libckks_hip.so for an in-box A/B (tools/ab_ntt.sh, tools/ab_inproc.py, tools/ab_engine_ops.sh):  tools/mkvariant.sh <name> [git-rev|-] [extra hipcc flags...]
#   rev "-" = the working tree.  Output: liberate_fhe_amd/csrc/variants/lib_<name>.so (git-ignored, travels with gpurun)
set -e
NAME=$1; REV=${2:--}; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/liberate_fhe_amd/csrc/variants; mkdir -p $OUT
SRC=$ROOT
if [ "$REV" != "-" ]; then
  SRC=$(mktemp -d /tmp/lfvar.XXXXXX)
  git -C $ROOT archive $REV liberate_fhe_amd/csrc include | tar -x -C $SRC
fi
C=$SRC/liberate_fhe_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o $OUT/lib_$NAME.so $C/ckks_hip.hip $C/ckks_ntt.hip $C/ckks_fused.hip $C/ckks_ks.hip $C/ckks_csprng.hip $C/ckks_ops.hip $C/ckks_w30.hip
[ "$REV" != "-" ] && rm -rf $SRC
echo built $OUT/lib_$NAME.so
