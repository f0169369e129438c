#!/bin/bash
# rocprofv3 kernel stats of ONE configuration at a given batch size (development aid): usage kstats_small.sh B BF16 TRAIN TAG
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kstats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BF16=$2 TRAIN=$3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $4 -- python3 $ROOT/scripts/quick_time.py $1 > $OUT/$4.log 2>&1
rm -f $OUT/$4_kernel_trace.csv
cut -d, -f1-4 $OUT/$4_kernel_stats.csv | head -30
