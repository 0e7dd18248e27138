#!/bin/bash
# SQ-level PMC passes over the bench kernels (run on the GPU box through gpurun)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
REPO=$PWD
ARGS="${BENCH_ARGS:---no-extra --steps 3 --warmup 1}"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq -o sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq2 -o sq2 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $OUT/pmc_sq3 -o sq3 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1
cd $REPO
python3 - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/pmc_sq*/*_results.db')):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size_x from counters_collection where kernel_name like '%ntt_%pass%' order by dispatch_id").fetchall()
    byd = {}
    for d, k, n, v, du, gx in rows:
        e = byd.setdefault(d, {}); e[n] = v; e['us'] = du/1e3; e['k'] = ('dp' if '<true>' in k else 'int'); e['grid'] = gx
    for d in sorted(byd)[-4:]: print(db.split('/')[-2], d, {k: (round(v,1) if isinstance(v,float) else v) for k,v in byd[d].items()})
PY
