# A/B of the bf16 weight-gradient phase: a launch per product (NERF_DW_BF16_MULTI=0) against all products in one launch (=1), same box.
# usage (GPU box): bash scripts/ab_dw_multi.sh > gpurun_out/ab_dw_multi.txt
set -e
export TRAIN=1 BF16=1
for B in ${BATCHES:-400 512 1024 2048 4096}; do
  for M in 0 1 0 1; do
    echo "== multi=$M B=$B"
    NERF_DW_BF16_MULTI=$M python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
