# A/B of launch shapes (development aid): variant libraries built with `make variant` / -DNERF_DW_GROUP_* / -DNERF_DWB_MIN_BLOCKS.
# usage: [BATCHES="512 2048"] [LIBS="a b"] [BF16=1] bash scripts/ab_dw_group.sh     ("-" = the shipped library)
set -e
for B in ${BATCHES:-512 2048}; do
  for L in ${LIBS:-- grp0}; do
    echo "== $L B=$B"
    if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
    NERF_HIP_LIB=$lib TRAIN=1 python scripts/quick_time.py $B 2>&1 | grep -v amdgpu.ids
  done
done
