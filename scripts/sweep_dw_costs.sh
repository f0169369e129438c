# tuning sweep (GPU box): shape factors of the multi-product bf16 weight-gradient launch; prints the weight-gradient phase per table
for B in 512 400; do
  for C in "170,150,135,107" "200,170,150,110" "230,200,160,115" "150,130,120,100" "200,250,135,107" "260,220,180,120" "170,150,160,125" "220,150,135,100"; do
    echo -n "B=$B costs=$C : "
    NERF_DW_BF16_COSTS=$C TRAIN=1 BF16=1 python scripts/quick_time.py $B 2>&1 | grep "bwd_dw" | sed -e "s/.*'bwd_dw': \([0-9.]*\)}/dw \1/"
  done
done
