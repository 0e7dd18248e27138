#!/bin/bash
# engine-op kernel traces only (gold + silver), summarized on the box
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/profile_engine_ops.sh
python3 tools/summarize_engine_ops.py r02 > /dev/null 2>&1
mkdir -p $OUT/profiles_out && cp profiles/r02_engine_ops_kernel_stats.txt profiles/r02_engine_ops_summary.json $OUT/profiles_out/
rm -rf $OUT/eo_*/
