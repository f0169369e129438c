// train_ops.hip -- the step-time floor around the hot path once the renderer is fast:
//   k_adam        fused Adam update of all 24 parameter tensors in ONE launch (torch.optim.Adam semantics,
//                 nerf.py:425: betas (0.9, 0.999), eps 1e-7, no weight decay / amsgrad)
//   k_gather_rays GPU-resident replacement of NeRFDataset.__getitem__ + DataLoader collation (loader.py:119-133):
//                 flat pixel index -> (row, column, pixel value, pose row, picture index) for a whole batch
#include "kernels.h"

namespace nerf {

__global__ __launch_bounds__(256) void k_adam(const AdamArgs a) {
  const int t = blockIdx.y;
  const int n = a.numel[t];
  float* __restrict__ p = a.param[t];
  const float* __restrict__ g = a.grad[t];
  float* __restrict__ m = a.m + a.offset[t];
  float* __restrict__ v = a.v + a.offset[t];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float gi = g[i];
    // exp_avg.lerp_(grad, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float mi = m[i] + (gi - m[i]) * a.one_minus_beta1;
    const float vi = v[i] * a.beta2 + (gi * gi) * a.one_minus_beta2;
    m[i] = mi;
    v[i] = vi;
    // denom = sqrt(v) / sqrt(1 - beta2^t) + eps;  param -= (lr / (1 - beta1^t)) * m / denom
    const float denom = sqrtf(vi) / a.bias2_sqrt + a.eps;
    p[i] = p[i] - a.step_size * (mi / denom);
  }
}

__global__ __launch_bounds__(256) void k_gather_rays(const GatherArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.B) return;
  const long long idx = a.index[i];            // flat pixel index in [0, n_pic * H * W)
  const long long hw = (long long)a.H * a.W;
  const int pic = (int)(idx / hw);
  const int rem = (int)(idx - (long long)pic * hw);
  // loader.py:119-126: pic_index = index // (H*W); row = rem // W; column = rem % W
  const int row = rem / a.W, col = rem - row * a.W;
  a.row[i] = row;
  a.col[i] = col;
  a.pic[i] = pic;
  const float* px = a.pixels + (size_t)idx * 3;
  a.pix_val[(size_t)i * 3 + 0] = px[0];
  a.pix_val[(size_t)i * 3 + 1] = px[1];
  a.pix_val[(size_t)i * 3 + 2] = px[2];
  const float* pr = a.poses + (size_t)pic * 17;
#pragma unroll
  for (int k = 0; k < 17; ++k) a.poses_bound[(size_t)i * 17 + k] = pr[k];
}

hipError_t launch_adam(const AdamArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_adam, dim3(64, 24), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_gather_rays(const GatherArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_rays, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace nerf
