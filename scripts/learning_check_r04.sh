# round-4 learning check (GPU box): the complete trainer (device sampler, NeRFModel.train_step, fused Adam, scheduler) on a multi-view-
# consistent scene with the round's final library: 4096-ray and 400-ray batches (the latter: multi-product weight gradients, fused ray
# stages, 4-wave workgroups, one-launch preparation), fp32 and bf16-MLP.  One JSON line per run.
for cfg in "2000 f32 1 4096" "2000 f32 2 4096" "1500 bf16 1 4096" "1500 bf16 2 4096" "6000 f32 1 400" "6000 bf16 1 400" "6000 bf16 2 400"; do
  echo "== $cfg"
  python scripts/teacher_student.py $cfg 2>&1 | grep -E "^\{|Error|error" | tail -1
done
