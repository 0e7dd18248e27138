#!/bin/bash
# Everything the committed profiles/ of a round are made from, in ONE GPU-box call:  tools/round_profiles.sh <tag>
#   PMC passes + kernel stats + bench line (pmc_round.sh), engine-op kernel traces (profile_engine_ops.sh), summaries.
# The summaries land in profiles/ on the box; they are copied to gpurun_out/profiles_<tag>/ to travel back.
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
bash tools/pmc_round.sh $TAG > gpurun_out/pmc_round_$TAG.log 2>&1
python3 tools/summarize_round.py $TAG > gpurun_out/summarize_round_$TAG.log 2>&1
bash tools/profile_engine_ops.sh > gpurun_out/profile_engine_ops_$TAG.log 2>&1
python3 tools/summarize_engine_ops.py $TAG > gpurun_out/summarize_engine_ops_$TAG.log 2>&1
# the line of record LAST: it folds in the summaries just written (they carry this build's source digest)
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 300 gpurun_out/bench_$TAG.json
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_bench_* profiles/${TAG}_engine_ops_* profiles/traffic_${TAG}.json gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/bench_$TAG.json gpurun_out/profiles_$TAG/${TAG}_bench_line.json 2>/dev/null
cp gpurun_out/*_$TAG.log gpurun_out/profiles_$TAG/ 2>/dev/null
rm -rf gpurun_out/pmc2_* gpurun_out/prof_* gpurun_out/eo_* gpurun_out/kt_* gpurun_out/kto   # raw traces stay on the box (64 MiB limit)
ls -la gpurun_out/profiles_$TAG
tail -3 gpurun_out/summarize_round_$TAG.log
