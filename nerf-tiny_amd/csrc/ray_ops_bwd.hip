// ray_ops_bwd.hip -- backward of the per-ray stages for MI355X (gfx950); one 64-lane wave per ray.
//   k_merge_bwd    d C_fine -> sorted-channel gradients (composite backward with reverse wave scans) ->
//                  un-sort through each channel's OWN permutation (quirk Q1) -> d rgb, d sigma of the coarse
//                  and fine samples and d t_fine (through the sorted deltas).
//   k_coarse_bwd   d t_fine -> inverse-CDF resampling backward (nerf.py:225-261; u, the ray-0 spacing and
//                  t_coarse carry no gradient) -> d w_coarse; plus d C_coarse; -> composite backward with the
//                  constant coarse delta -> d sigma_coarse, d rgb_coarse.
// Formulas: SURVEY.md 8a BWD (verified there against autograd in fp64).  These replace the autograd graph the
// reference builds for nerf.py:286-323 when loss.backward() runs (nerf.py:473).
#include "kernels.h"
#include "ray_parts_bwd.h"

namespace nerf {

// ---------------------------------------------------------------------------------------------
// k_merge_bwd: grid = B blocks of 64 threads; dynamic LDS = 4 * N floats
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_merge_bwd(const MergeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  merge_bwd_ray(a, blockIdx.x, threadIdx.x, sm_f, [] { __syncthreads(); });
}

// ---------------------------------------------------------------------------------------------
// k_coarse_bwd: one wave per ray, 4 rays per 256-thread block
// ---------------------------------------------------------------------------------------------
// dynamic LDS: 4 waves x (5 * Nc floats) then 4 waves x (Nf uint16, padded to even)
__global__ __launch_bounds__(256) void k_coarse_bwd(const CoarseBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm_c[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * 4 + wv;
  const bool live = ray_raw < a.B;
  const int ray = live ? ray_raw : a.B - 1;
  float* w = sm_c + (size_t)wv * 5 * a.Nc;
  uint16_t* ks = reinterpret_cast<uint16_t*>(sm_c + (size_t)4 * 5 * a.Nc) + (size_t)wv * ((a.Nf + 1) & ~1);
  coarse_bwd_ray(a, ray, live, lane, w, ks, [] { __syncthreads(); });
}

// ---------------------------------------------------------------------------------------------
size_t merge_bwd_lds_bytes(int N) { return (size_t)4 * N * sizeof(float); }
hipError_t launch_merge_bwd(const MergeBwdArgs& a, hipStream_t st) {
  const size_t lds = merge_bwd_lds_bytes(a.Nc + a.Nf);
  hipLaunchKernelGGL(k_merge_bwd, dim3(a.B), dim3(64), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_coarse_bwd(const CoarseBwdArgs& a, hipStream_t st) {
  const size_t lds = (size_t)4 * 5 * a.Nc * sizeof(float) + (size_t)4 * ((a.Nf + 1) & ~1) * sizeof(uint16_t);
  if (lds > 64 * 1024)
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
  hipLaunchKernelGGL(k_coarse_bwd, dim3((a.B + 3) / 4), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace nerf
