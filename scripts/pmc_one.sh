#!/bin/bash
# PMC passes of ONE bench leg (development aid): usage pmc_one.sh NAME "bench args"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; args=$2
mkdir -p $ROOT/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  ptag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc -o ${name}_$ptag -- python3 $ROOT/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $ROOT/gpurun_out/pmc/${name}_$ptag.err || { echo "pmc pass $name $ptag failed"; tail -3 $ROOT/gpurun_out/pmc/${name}_$ptag.err; }
done
rm -f $ROOT/gpurun_out/pmc/*_kernel_trace.csv $ROOT/gpurun_out/pmc/*agent_info.csv
echo "pmc $name done"
