# how much store-latency tolerance is worth in the bf16 training kernels (development aid): the shipped library against variants with
# shallower LDS rings (forward 8 -> 6 -> 4 slots, chain 5 -> 4 -> 3) and against the aliasing build (saves hit L2: no write drain)
#   make -C nerf-tiny_amd/csrc variant NAME=ns6b4 DEFS="-DNERF_BF_NS=6 -DNERF_BB_NS=4"; ... NAME=ns4b3 ...; ... NAME=alias DEFS=-DNERF_TIMING_SAVE_ALIAS
for L in ${LIBS:-- ns6b4 ns4b3 alias}; do
  if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
  for rep in 1 2; do
    echo "== $L"
    NERF_HIP_LIB=$lib BF16=1 TRAIN=1 python scripts/quick_time.py ${B:-4096} 2>&1 | grep -v amdgpu.ids
  done
done
