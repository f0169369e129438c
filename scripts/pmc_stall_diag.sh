#!/bin/bash
# Stall attribution of ONE bench leg with rocprofv3 PMC passes (development aid; run through gpurun):
#     bash scripts/pmc_stall_diag.sh NAME "bench args"        e.g.  train_bf16 "--mode train --mlp bf16"
# Replaces round 4's uncommitted script of the same name, whose third pass never returned (gpurun_out/r04s_call.txt: killed after 420 s of
# silence; profiles/README.md "The silence-killed run of round 4").  Rules this one keeps:
#   * ONE progress line per pass on STDOUT (never only into a redirected file), before AND after the pass;
#   * `python3 bench.py ...` stands directly after `--` (no env / bash -c / taskset hop: the profiler's preloaded library has initialised
#     the GPU by then, and any such hop is an exec from a GPU-initialised process);
#   * --pmc only together with --kernel-trace (never --sys-trace / --runtime-trace / hip / hsa trace domains);
#   * every pass under its own `timeout -k 10`, and after a pass that times out NO further GPU step is started (exit 3);
#   * per pass <= 8 SQ counters, <= 2 GRBM counters, TCC counters alone (MI355X_MICROARCH.md "rocprofv3 PMC slots"); no SQ_*_LEVEL_* /
#     SQ_ACCUM_PREV* counters (level counters need an accumulate pair in the following slot; they are what this script no longer asks for).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
name=${1:?usage: pmc_stall_diag.sh NAME "bench args"}; args=$2
OUT=$ROOT/gpurun_out/stall
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASS_TIMEOUT=${PASS_TIMEOUT:-240}
passes=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE"
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
)
p=0
for pass in "${passes[@]}"; do
  p=$((p + 1))
  echo "[pmc_stall_diag] $name pass $p/${#passes[@]} start: $pass"
  timeout -k 10 $PASS_TIMEOUT rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o ${name}_p$p -- python3 $ROOT/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${name}_p$p.bench.json 2> $OUT/${name}_p$p.err
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "[pmc_stall_diag] $name pass $p TIMED OUT after ${PASS_TIMEOUT}s (rc $rc): counters were: $pass -- stopping, no further GPU step"
    tail -5 $OUT/${name}_p$p.err
    exit 3
  fi
  if [ $rc -ne 0 ]; then echo "[pmc_stall_diag] $name pass $p failed (rc $rc)"; tail -5 $OUT/${name}_p$p.err; fi
  echo "[pmc_stall_diag] $name pass $p done (rc $rc)"
done
rm -f $OUT/*agent_info.csv
python3 $ROOT/scripts/summarize_stall.py $OUT $name ${#passes[@]}
echo "[pmc_stall_diag] $name complete"
