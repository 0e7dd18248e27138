#!/bin/bash
# final validation + every profile of the round on one box; the summaries (not the databases: gpurun merges at most
# 64 MiB back) come home in gpurun_out/profiles_out/
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/r02_pmc.sh | tail -2
bash tools/profile_engine_ops.sh
python3 tools/class_split.py 2>/dev/null > $OUT/class_split.txt; cat $OUT/class_split.txt
cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/pmc2_class -o p -- python3 $OLDPWD/tools/class_split.py > /dev/null 2>&1; cd $OLDPWD
python3 tools/pmc_report.py "gpurun_out/pmc2_class/*_results.db" > $OUT/class_split_pmc.txt 2>&1
python3 tools/summarize_r02.py > /dev/null 2>&1
python3 tools/summarize_engine_ops.py r02 > /dev/null 2>&1
mkdir -p $OUT/profiles_out && cp profiles/r02_bench_kernel_stats.txt profiles/r02_bench_pmc.txt profiles/traffic_r02.json profiles/r02_engine_ops_pmc.txt profiles/r02_engine_ops_kernel_stats.txt profiles/r02_engine_ops_summary.json $OUT/profiles_out/
rm -rf $OUT/pmc2_* $OUT/eo_*/ $OUT/prof_r02
du -sh $OUT
