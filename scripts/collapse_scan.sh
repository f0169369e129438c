#!/bin/bash
# Learning-stability scan (development aid): teacher/student runs over several seeds, both MLP precisions; one summary line per run.
# usage: bash scripts/collapse_scan.sh ITERS "SEEDS" "MODES"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
IT=${1:-3000}; SEEDS=${2:-"1 2 3 11 12"}; MODES=${3:-"bf16 f32"}
mkdir -p $ROOT/gpurun_out/scan
for m in $MODES; do for s in $SEEDS; do
  python $ROOT/scripts/teacher_student.py $IT $m $s > $ROOT/gpurun_out/scan/${m}_${s}.log 2>&1
  echo "$m seed $s: $(grep -o 'LOSS\] [0-9.]*' $ROOT/gpurun_out/scan/${m}_${s}.log | cut -d' ' -f2 | tr '\n' ' ') | $(tail -1 $ROOT/gpurun_out/scan/${m}_${s}.log | grep -o '"psnr_train_views_after_db": [0-9.]*')"
done; done
