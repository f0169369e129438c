# round-4 A/B (GPU box): 4-wave workgroups for small passes of the bf16 training kernels
set -e
for B in 400 512 1024; do
  for V in 0 1 0 1; do
    echo "== train bf16 B=$B four_wave=$V"
    NERF_BF16_4WAVE=$V TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
