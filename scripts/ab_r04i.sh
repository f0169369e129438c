# round-4 A/B (GPU box): grouping of the small-block bf16 weight-gradient products at LARGE batches
set -e
for B in 4096 2048; do
  for V in 0 1 2 0 1 2; do
    echo "== train bf16 B=$B smallgroup=$V"
    NERF_DW_BF16_SMALLGROUP=$V TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
