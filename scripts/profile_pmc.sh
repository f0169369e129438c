#!/bin/bash
# PMC passes for the forward bench on the GPU box (run through gpurun).  Counters are collected in separate passes
# (TCC slot limits; MI355X_MICROARCH.md "rocprofv3 PMC slots") and only together with --kernel-trace.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${1:-forward}
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o ${MODE}_$tag -- python3 $ROOT/bench.py --mode $MODE --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${MODE}_$tag.bench.json 2> $OUT/${MODE}_$tag.err || { echo "pass $tag failed"; tail -5 $OUT/${MODE}_$tag.err; }
  echo "pass $tag done"
done
ls $OUT
