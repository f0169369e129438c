#!/bin/bash
# Round-end evidence: bench lines, rocprofv3 kernel stats and PMC passes (run through gpurun; results under gpurun_out/final,
# summarised into profiles/ by scripts/summarize_pmc.py and copied by hand).  The program sits directly after `--`.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 50 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench (default line: forward fp32 + extra legs + cpu baseline) done"
python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_train.json 2> $OUT/bench_train.err && echo "bench train done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/fwd_under_rocprof.json 2> $OUT/fwd_rocprof.err && echo "rocprof fwd done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $ROOT/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $OUT/train_under_rocprof.json 2> $OUT/train_rocprof.err && echo "rocprof train done"
rm -f $OUT/*_kernel_trace.csv
bash $ROOT/scripts/profile_pmc.sh forward forward "--no-extra" > $OUT/pmc_fwd.log 2>&1 && echo "pmc forward done"
bash $ROOT/scripts/profile_pmc.sh train train_f32 "--no-extra" > $OUT/pmc_train.log 2>&1 && echo "pmc train done"
ls $OUT
