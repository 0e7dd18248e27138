#!/bin/bash
# PMC passes (separate runs, as the guide prescribes) over the bench step and over gold / silver cc_mult:
#   tools/pmc_round.sh <tag>      (on the GPU box; then tools/summarize_round.py <tag> -> profiles/)
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
# build ONCE, unprofiled: a profiled process must never spawn the compiler (the profiler's preload would ride along into hipcc,
# and the minutes of compilation would sit inside a counter run)
python3 -c 'import __graft_entry__ as g; g.build()' > $OUT/build_$TAG.log 2>&1 || { echo "build failed"; tail -5 $OUT/build_$TAG.log; exit 1; }
run() {  # tag, counters..., then "--", then program args
  local TAG=$1; shift; local CNT=(); while [ "$1" != "--" ]; do CNT+=("$1"); shift; done; shift
  cd /tmp && rocprofv3 --pmc "${CNT[@]}" --kernel-trace -d $OUT/pmc2_$TAG -o p -- python3 "$@" > $OUT/pmc2_$TAG.log 2>&1; cd $REPO
}
B="$REPO/bench.py --no-extra --steps 5 --warmup 2"
run bench_fetch FETCH_SIZE -- $B
run bench_write WRITE_SIZE -- $B
run bench_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -- $B
for P in gold silver; do
  for OP in cc_mult rotate; do
    C="$REPO/tools/ccmult_profile.py $P $OP --mark"
    T=$P; [ $OP = rotate ] && T=${P}_rot
    run ${T}_fetch FETCH_SIZE -- $C
    run ${T}_write WRITE_SIZE -- $C
    run ${T}_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -- $C
  done
done
# batched groups under one key (nct = 4 per key-switch launch set), gold: the traffic per ciphertext of BASELINE configs[4]
for OP in rotate_batch cc_mult_batch; do
  C="$REPO/tools/ccmult_profile.py gold $OP --mark"
  T=gold_$OP
  run ${T}_fetch FETCH_SIZE -- $C
  run ${T}_write WRITE_SIZE -- $C
  run ${T}_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -- $C
done
# kernel stats of the bench command itself (the file the roofline numbers are checked against)
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o stats -- python3 $REPO/bench.py --no-extra > $OUT/prof_$TAG.log 2>&1; cd $REPO
python3 bench.py --no-extra > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -c 400 $OUT/bench_$TAG.json   # (round_profiles.sh replaces it by the full line)
ls $OUT | grep pmc2 | head -20
