#!/bin/bash
# PMC passes for the forward bench on the GPU box (run through gpurun).  Counters are collected in separate passes
# (TCC slot limits; MI355X_MICROARCH.md "rocprofv3 PMC slots") and only together with --kernel-trace.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${1:-forward}
TAG=${2:-$MODE}          # file prefix (summarize_pmc.py's "mode" argument)
EXTRA=${3:-}             # extra bench.py arguments, e.g. "--mlp bf16"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o ${TAG}_$tag -- python3 $ROOT/bench.py --mode $MODE $EXTRA --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_$tag.bench.json 2> $OUT/${TAG}_$tag.err || { echo "pass $tag failed"; tail -5 $OUT/${TAG}_$tag.err; }
  echo "pass $tag done"
done
ls $OUT
