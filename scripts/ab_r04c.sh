# round-4 A/B (GPU box): one-launch preparation and cost-weighted workgroup allocation of the bf16 weight-gradient launch
set -e
for B in 400 512; do
  for V in "NERF_PREP_BF16=0 NERF_DW_BF16_COST=0" "NERF_PREP_BF16=1 NERF_DW_BF16_COST=0" "NERF_PREP_BF16=1 NERF_DW_BF16_COST=1" "NERF_PREP_BF16=0 NERF_DW_BF16_COST=0" "NERF_PREP_BF16=1 NERF_DW_BF16_COST=1"; do
    echo "== train bf16 B=$B $V"
    env $V TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
  for V in "NERF_PREP_BF16=0" "NERF_PREP_BF16=1" "NERF_PREP_BF16=0" "NERF_PREP_BF16=1"; do
    echo "== forward bf16 B=$B $V"
    env $V TRAIN=0 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
