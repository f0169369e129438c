# round-4 A/B (GPU box): per-ray stages fused into the field launches (small bf16 training batches)
set -e
for B in 400 512; do
  for V in 0 1 0 1; do
    echo "== train bf16 B=$B fuse_rays=$V"
    NERF_FUSE_RAYS=$V TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
