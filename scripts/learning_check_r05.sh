# round-5 learning check (GPU box): the complete trainer on a multi-view-consistent scene, same seeds, exact fp32 vs the opt-in split-fp32
# train step vs the bf16-MLP variant; one JSON line per run (progress line per run on stdout).
for cfg in "2000 f32 1 4096" "2000 split 1 4096" "2000 f32 2 4096" "2000 split 2 4096" "1500 bf16 1 4096" "6000 f32 1 400" "6000 split 1 400"; do
  echo "== $cfg"
  python scripts/teacher_student.py $cfg 2>&1 | grep -E "^\{|Error|error" | tail -1
done
