# round 5 (GPU box): how often does a run die (the reference's exit(0) condition: DESIGN.md section 6) -- exact fp32 vs the split-fp32 train step,
# same seeds, 1,000 iterations of 4,096 rays at lr 3e-4.  One line per run.
for seed in 3 4 5 6 7 8 9 10; do
  for mode in f32 split; do
    echo -n "seed $seed $mode: "
    python scripts/teacher_student.py 1000 $mode $seed 4096 2>&1 | grep -E "^\{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['psnr_heldout_before_db'], '->', d['psnr_heldout_after_db'], 'train views', d['psnr_train_views_after_db'], 'rays/s', d['trainer_rays_per_s'])"
  done
done
