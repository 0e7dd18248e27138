#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM-traffic PMC passes.
# Outputs land in gpurun_out/; copy the summaries you want judged into profiles/.
set -u
R=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/bench_$R.json 2> $OUT/bench_$R.err
tail -c 3000 $OUT/bench_$R.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$R -o stats -- python3 $OLDPWD/bench.py --no-extra > $OUT/prof_$R.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$R -o fetch -- python3 $OLDPWD/bench.py --no-extra --steps 5 --warmup 2 > $OUT/pmc_fetch_$R.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$R -o write -- python3 $OLDPWD/bench.py --no-extra --steps 5 --warmup 2 > $OUT/pmc_write_$R.log 2>&1
cd $OLDPWD
find $OUT -name "*.csv" | head -20
