#!/bin/bash
# Evidence for the bf16-MLP variant: bench lines and rocprofv3 kernel stats (run through gpurun; results under gpurun_out/final_bf16).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final_bf16
mkdir -p $OUT
cd $ROOT
python3 bench.py --mlp bf16 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_fwd.json 2> $OUT/bench_fwd.err && echo "bench fwd done"
python3 bench.py --mlp bf16 --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err && echo "bench train done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python3 $ROOT/bench.py --mlp bf16 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/fwd_under_rocprof.json 2> $OUT/fwd_rocprof.err && echo "rocprof fwd done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $ROOT/bench.py --mlp bf16 --mode train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/train_under_rocprof.json 2> $OUT/train_rocprof.err && echo "rocprof train done"
rm -f $OUT/*_kernel_trace.csv
ls $OUT
