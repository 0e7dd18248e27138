#!/bin/bash
# kernel traces of cc_mult / rotate_single at silver and gold (run on the GPU box): tools/profile_engine_ops.sh
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
python3 -c 'import __graft_entry__ as g; g.build()' > $OUT/build_eo.log 2>&1 || { echo "build failed"; exit 1; }   # never under the profiler
for P in silver gold; do for OP in cc_mult rotate; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/eo_${P}_${OP} -o kt -- python3 $REPO/tools/ccmult_profile.py $P $OP --mark > $OUT/eo_${P}_${OP}.log 2>&1
  cd $REPO; tail -1 $OUT/eo_${P}_${OP}.log
done; done
