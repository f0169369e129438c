# A/B of variant libraries against the shipped one on the same box (development aid).  Variants: `make -C nerf-tiny_amd/csrc variant
# NAME=x DEFS="-D..."` -> nerf-tiny_amd/libnerf_hip_x.so, e.g. -DNERF_DW_GROUP_MAX_ROWS=0 (a launch per weight-gradient product),
# -DNERF_BX_GROUPS=4 (two-column bf16 inference kernel), -DNERF_TIMING_SAVE_ALIAS.
# usage: [BATCHES="512 4096"] [LIBS="- x"] [BF16=1] [TRAIN=0] bash scripts/ab_variants.sh     ("-" = the shipped library)
set -e
export TRAIN=${TRAIN:-1}
for B in ${BATCHES:-512 4096}; do
  for L in ${LIBS:--}; do
    echo "== $L B=$B"
    if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
    NERF_HIP_LIB=$lib python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
