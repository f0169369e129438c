#!/bin/bash
# Copies what scripts/collect_profiles_r05.sh left under gpurun_out/<TAG> (+ gpurun_out/pmc) into profiles/r05_* (run in the build
# container after the gpurun call).   usage: bash scripts/publish_profiles_r05.sh [TAG]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r05}
SRC=$ROOT/gpurun_out/$TAG
cp $SRC/bench_default.json $ROOT/profiles/r05_bench_default.json
cp $SRC/bench_extra.json $ROOT/profiles/r05_bench_extra.json
for leg in fwd_f32 train_f32 fwd_bf16 train_bf16 split_fwd train_split; do
  cp $SRC/${leg}_kernel_stats.csv $ROOT/profiles/r05_${leg}_kernel_stats.csv
  cp $SRC/${leg}_under_rocprof.json $ROOT/profiles/r05_${leg}_bench_under_rocprof.json
  python3 $ROOT/scripts/summarize_pmc.py r05 $leg > /dev/null
done
for n in train_bf16_b512 fwd_bf16_b512 train_bf16_b400 train_f32_b400; do
  cp $SRC/${n}_kernel_stats.csv $ROOT/profiles/r05_${n}_kernel_stats.csv
done
ls -la $ROOT/profiles | grep r05_
