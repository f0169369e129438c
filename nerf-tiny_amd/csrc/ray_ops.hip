// ray_ops.hip -- the per-ray (HBM/latency-bound) stages of the hot path for MI355X (gfx950):
//   k_rays              pixel -> camera/world direction, coarse depths, per-ray direction-branch vector
//   k_coarse            sigma -> weights (inclusive transmittance), C_coarse, inverse-CDF resampling
//   k_merge             merge coarse+fine, five independent channel sorts, composite -> C_fine
//   k_ray_loss          sum-of-squares loss and its gradient
// One 64-lane wave owns one ray: the transmittance cumsum / CDF are wave scans (fp64 accumulator like
// ATen's CPU cumsum, rounded to fp32 per element), searchsorted is a per-lane binary search in LDS and
// the channel sorts are bitonic networks in LDS.  Built with -ffp-contract=off: the arithmetic order is
// the reference's (SURVEY.md 8a SPEC); fused multiply-adds appear only where written as fmaf.
#include "kernels.h"
#include "prep_parts.h"
#include "ray_parts.h"

namespace nerf {

// ---------------------------------------------------------------------------------------------
// k_rays: nerf.py:52-67 (pose split), 186-197 (pixel -> unit camera dir), 211 (world dir), 288 (coarse
// depths, numpy.linspace in fp32) and the gamma_d half of dir_info (nerf.py:118) which is constant per ray.
// grid = B blocks of 128 threads.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(128) void k_rays(const RaysArgs a) {
  __shared__ float gd[DIR_DIM];
  ray_block(a, blockIdx.x, threadIdx.x, gd);
}

// ---------------------------------------------------------------------------------------------
// k_coarse: get_density (nerf.py:263-272) with delta = (far-near)/Nc, color_cum (nerf.py:274-281, :320),
// resample (nerf.py:225-261).  One wave per ray, 4 rays per 256-thread block.
// ---------------------------------------------------------------------------------------------


__global__ __launch_bounds__(256) void k_coarse(const CoarseArgs a) {
  __shared__ float s_w[4][MAXN];
  __shared__ float s_cdf[4][MAXN];
  __shared__ float s_t[4][MAXN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  coarse_ray_stage(a, blockIdx.x * 4 + wv, lane, s_w[wv], s_cdf[wv], s_t[wv], [] { __syncthreads(); });
}

// ---------------------------------------------------------------------------------------------
// k_merge: nerf.py:302-321.  cat -> sort(dim=1) of a [B,N,5] bundle = five INDEPENDENT ascending channel
// sorts (quirk Q1), delta_i = t_{i+1} - t_i with the last = `last`, weights, C_fine.
// One wave (64-thread block) per ray; bitonic sort over P = next_pow2(N) slots per channel in LDS.
// Ties are broken by original index (a stable sort); indices are kept for the backward pass.
// ---------------------------------------------------------------------------------------------

template <bool WITH_IDX>  // WITH_IDX: carry the original index (stable order + permutation for backward)
__global__ __launch_bounds__(64) void k_merge(const MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* val = reinterpret_cast<float*>(smem_raw);                      // [5][P]
  uint16_t* idx = reinterpret_cast<uint16_t*>(val + 5 * (size_t)a.P);  // [5][P]
  merge_ray_stage<WITH_IDX>(a, blockIdx.x, threadIdx.x, val, idx, [] { __syncthreads(); });
}

// ---------------------------------------------------------------------------------------------
// k_ray_loss: nerf.py:325-331 (sum, not mean) and d loss / dC.  Single 1024-thread block (B*3 elements).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ray_loss(const float* Cc, const float* Cf, const float* Ct, int n, float* loss, float* dCc,
                                                   float* dCf) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    float d1, d2, term;
    ray_loss_element(Cc[i], Cf[i], Ct[i], d1, d2, term);
    acc += term;
    if (dCc) dCc[i] = d1;
    if (dCf) dCf[i] = d2;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    loss[0] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_rays(const RaysArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_rays, dim3(a.B), dim3(128), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_coarse(const CoarseArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_coarse, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  return hipGetLastError();
}
size_t merge_lds_bytes(int P) { return (size_t)5 * P * (sizeof(float) + sizeof(uint16_t)); }
hipError_t launch_merge(const MergeArgs& a, hipStream_t st) {
  const size_t lds = merge_lds_bytes(a.P);
  if (a.perm || a.joint) {  // (the joint-sort mode needs the depth channel's permutation, stored or not)
    if (lds > 48 * 1024)
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    hipLaunchKernelGGL(k_merge<true>, dim3(a.B), dim3(64), lds, st, a);
  } else {
    if (lds > 48 * 1024)
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    hipLaunchKernelGGL(k_merge<false>, dim3(a.B), dim3(64), lds, st, a);
  }
  return hipGetLastError();
}
hipError_t launch_ray_loss(const float* Cc, const float* Cf, const float* Ct, int B, float* loss, float* dCc, float* dCf, hipStream_t st) {
  hipLaunchKernelGGL(k_ray_loss, dim3(1), dim3(1024), 0, st, Cc, Cf, Ct, B * 3, loss, dCc, dCf);
  return hipGetLastError();
}

}  // namespace nerf
