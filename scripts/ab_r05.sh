# A/B of variant libraries on one box (development aid; each line = scripts/quick_time.py: per-kernel ms of one step + the step time).
#   usage: LIBS="- half alt" CFGS="1:1 0:1" BATCHES="4096" bash scripts/ab_r05.sh     (CFGS entries are BF16:TRAIN; "-" = the shipped library)
for B in ${BATCHES:-4096}; do
  for cfg in ${CFGS:-1:1}; do
    for rep in 1 2; do
      for L in ${LIBS:--}; do
        if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
        echo "== $L B=$B bf16=${cfg%%:*} train=${cfg##*:}"
        NERF_HIP_LIB=$lib BF16=${cfg%%:*} TRAIN=${cfg##*:} python scripts/quick_time.py $B 2>&1 | tail -2
      done
    done
  done
done
