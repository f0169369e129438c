#!/bin/bash
# Round-5 evidence in ONE gpurun call: the driver's bench line, rocprofv3 kernel stats and PMC passes (separate passes per counter group,
# only with --kernel-trace) for all four legs of the path: forward / train step x fp32 / bf16-MLP.  Results under gpurun_out/<TAG>;
# scripts/summarize_pmc.py reduces the PMC CSVs into profiles/ (run afterwards in the build container).  The program sits directly
# after `--`.   usage: bash scripts/collect_profiles_r05.sh [TAG]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $ROOT/gpurun_out/pmc
cd $ROOT
echo "[collect] bench default line"
python3 bench.py --side-file $OUT/bench_extra.json > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench (default line) done: $(wc -c < $OUT/bench_default.json) bytes"
cd /tmp && export TMPDIR=/tmp
for leg in "fwd_f32:--mode forward --mlp f32 --steps 20 --warmup 3" "train_f32:--mode train --mlp f32 --steps 6 --warmup 2" "fwd_bf16:--mode forward --mlp bf16 --steps 40 --warmup 5" "train_bf16:--mode train --mlp bf16 --steps 10 --warmup 3" "split_fwd:--mode forward --split --steps 30 --warmup 4" "train_split:--mode train --split --steps 10 --warmup 3"; do
  name=${leg%%:*}; args=${leg#*:}
  echo "[collect] $name: kernel stats"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $name -- python3 $ROOT/bench.py $args --no-cpu-baseline --no-extra --side-file $OUT/${name}_extra.json > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.err && echo "rocprof stats $name done"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
    ptag=$(echo $pass | cut -d' ' -f1)
    echo "[collect] $name: pmc pass $ptag"
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc -o ${name}_$ptag -- python3 $ROOT/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-extra --side-file $OUT/${name}_${ptag}_extra.json > $OUT/${name}_$ptag.bench.json 2> $OUT/${name}_$ptag.err || { echo "pmc pass $name $ptag failed"; tail -3 $OUT/${name}_$ptag.err; }
  done
  echo "pmc $name done"
done
# the per-rank share of an 8-GPU strong-scaling step (512 rays) and the reference's own batch (400 rays): kernel stats of the bf16 legs
# (the one-launch preparation, the multi-product weight-gradient launch, the ray-pair inference kernel live here)
for cfg in "512:1:1:train_bf16_b512" "512:1:0:fwd_bf16_b512" "400:1:1:train_bf16_b400" "400:0:1:train_f32_b400"; do
  IFS=: read B BF TR NAME <<< "$cfg"
  BF16=$BF TRAIN=$TR rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $NAME -- python3 $ROOT/scripts/quick_time.py $B > $OUT/$NAME.log 2>&1 && echo "kstats $NAME done"
done
rm -f $OUT/*_extra.json.tmp $OUT/*_kernel_trace.csv $ROOT/gpurun_out/pmc/*_kernel_trace.csv $ROOT/gpurun_out/pmc/*agent_info.csv
ls $OUT | head -50
