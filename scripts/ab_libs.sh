# A/B of variant libraries on one box (development aid): forward and train step, bf16 and split, at the given batch sizes.
#   usage: LIBS="- nt" BATCHES="4096 512" bash scripts/ab_libs.sh      ("-" = the shipped library; x = nerf-tiny_amd/libnerf_hip_x.so)
for B in ${BATCHES:-4096}; do
  for L in ${LIBS:--}; do
    if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
    for rep in 1 2; do
      for cfg in "1 1 0" "1 0 0" "0 0 1"; do
        set -- $cfg
        echo "== $L B=$B bf16=$1 train=$2 split=$3"
        NERF_HIP_LIB=$lib BF16=$1 TRAIN=$2 SPLIT=$3 python scripts/quick_time.py $B 2>&1 | grep "^B="
      done
    done
  done
done
