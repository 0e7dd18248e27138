#!/bin/bash
set -u
export TMPDIR=/tmp
V=liberate_fhe_amd/csrc/variants
LF_HIP_LIB=$PWD/$V/lib_duo4.so timeout 900 python -m pytest tests/test_ntt_cuda_gpu.py -m gpu -x -q 2>&1 | tail -3
tools/r02_ab4.sh $V/lib_old.so $V/lib_u1.so $V/lib_u1w4.so $V/lib_duo2.so $V/lib_duo2_notile.so $V/lib_duo4.so
