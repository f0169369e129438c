# round-4 A/B (GPU box): one-launch preparation (flag words), ray-pair inference kernel, bf16 train step
set -e
for B in 400 512 1024; do
  for V in "NERF_PREP_BF16=0 NERF_PAIR_BF16=0" "NERF_PREP_BF16=1 NERF_PAIR_BF16=0" "NERF_PREP_BF16=1 NERF_PAIR_BF16=1" "NERF_PREP_BF16=0 NERF_PAIR_BF16=0" "NERF_PREP_BF16=1 NERF_PAIR_BF16=1"; do
    echo "== forward bf16 B=$B $V"
    env $V TRAIN=0 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
for B in 400 512; do
  for V in "NERF_PREP_BF16=0" "NERF_PREP_BF16=1" "NERF_PREP_BF16=0" "NERF_PREP_BF16=1"; do
    echo "== train bf16 B=$B $V"
    env $V TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
echo "== forward bf16 B=4096 (the separate kernels: unchanged?)"
TRAIN=0 BF16=1 python scripts/quick_time.py 4096 2>&1 | grep -v amdgpu.ids
