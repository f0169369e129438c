#!/bin/bash
# Copies what scripts/collect_profiles_r03.sh left under gpurun_out/<TAG> (+ gpurun_out/pmc) into profiles/r03_* (run in the build
# container after the gpurun call).   usage: bash scripts/publish_profiles_r03.sh [TAG]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r03}
SRC=$ROOT/gpurun_out/$TAG
cp $SRC/bench_default.json $ROOT/profiles/r03_bench_default.json
for leg in fwd_f32 train_f32 fwd_bf16 train_bf16 split_fwd; do
  cp $SRC/${leg}_kernel_stats.csv $ROOT/profiles/r03_${leg}_kernel_stats.csv
  cp $SRC/${leg}_under_rocprof.json $ROOT/profiles/r03_${leg}_bench_under_rocprof.json
  python3 $ROOT/scripts/summarize_pmc.py r03 $leg > /dev/null
done
ls -la $ROOT/profiles | grep r03_
