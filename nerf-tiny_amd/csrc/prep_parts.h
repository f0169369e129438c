// prep_parts.h -- the per-call preparation steps as device functions, shared by their stand-alone kernels (k_fold_weights in
// field_fwd.hip, k_rays in ray_ops.hip) and by the ONE-launch preparation of the bf16 paths (prep_bf16.hip).  MI355X / gfx950 only.
#pragma once
#include "kernels.h"

namespace nerf {

// The fold, in front of every packer (fp32 and bf16-MLP variant alike): fold[0 .. 128) = b_fold = W_dir[:, 24:] b_pi (k_rays adds it to
// every ray's dir_info start vector), fold[128 + o * 256 + k] = W_fold[o][k] = sum_j W_dir[o][24 + j] * W_pi[j][k] (common.h SEG_FOLD).
// 129 blocks of 256 threads: block o < 128 = row o of W_fold, thread (part, k4): four fp32 fma chains over a quarter of the j range each
// (the W_dir element of a step is wave-uniform, the W_pi row a coalesced KiB), the quarters added in a fixed order; block 128 = b_fold.
__device__ __forceinline__ void fold_block(const Weights24& w, float* __restrict__ fold, const int block /* 0..HALF */) {
  __shared__ float4 part_sum[3][64];
  const int t = threadIdx.x;
  if (block == HALF) {
    __shared__ float bsum[HALF];
    const int o = t & (HALF - 1), half = t >> 7;
    const float* dr = w.p[W_DIR] + (size_t)o * (WIDTH + DIR_DIM) + DIR_DIM + half * (WIDTH / 2);
    const float* bp = w.p[B_PI] + half * (WIDTH / 2);
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < WIDTH / 2; ++j) s = __builtin_fmaf(dr[j], bp[j], s);
    if (half) bsum[o] = s;
    __syncthreads();
    if (!half) fold[o] = s + bsum[o];
    return;
  }
  const int o = block, k0 = 4 * (t & 63);
  const int part = __builtin_amdgcn_readfirstlane(t >> 6);
  const float* dr = w.p[W_DIR] + (size_t)o * (WIDTH + DIR_DIM) + DIR_DIM + part * (WIDTH / 4);
  const float* pc = w.p[W_PI] + (size_t)part * (WIDTH / 4) * WIDTH + k0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int j = 0; j < WIDTH / 4; ++j) {
    const float d = dr[j];
    const float4 q = *reinterpret_cast<const float4*>(pc + (size_t)j * WIDTH);
    acc.x = __builtin_fmaf(d, q.x, acc.x);
    acc.y = __builtin_fmaf(d, q.y, acc.y);
    acc.z = __builtin_fmaf(d, q.z, acc.z);
    acc.w = __builtin_fmaf(d, q.w, acc.w);
  }
  if (part) part_sum[part - 1][t & 63] = acc;
  __syncthreads();
  if (part == 0) {
    const float4 p1 = part_sum[0][t], p2 = part_sum[1][t], p3 = part_sum[2][t];
    acc.x = (acc.x + p1.x) + (p2.x + p3.x);
    acc.y = (acc.y + p1.y) + (p2.y + p3.y);
    acc.z = (acc.z + p1.z) + (p2.z + p3.z);
    acc.w = (acc.w + p1.w) + (p2.w + p3.w);
    *reinterpret_cast<float4*>(fold + HALF + (size_t)o * WIDTH + k0) = acc;
  }
}

// k_rays: nerf.py:52-67 (pose split), 186-197 (pixel -> unit camera dir), 211 (world dir), 288 (coarse
// depths, numpy.linspace in fp32) and the gamma_d half of dir_info (nerf.py:118) which is constant per ray.
// The ray record of one ray in registers (every calling lane computes the same values): rec[0..8] R row-major, [9..11] o, [12..14] d_cam,
// [15..17] d_wrd, [18] near, [19] far, [20] (far - near)/(Nc - 1), [21] (far - near)/Nc (common.h RF_*).
__device__ __forceinline__ void ray_record(const RaysArgs& a, const int ray, float (&rec)[RAYF]) {
  const float* pb = a.pb + (size_t)ray * 17;
  // x <- row, y <- column (quirk Q2)
  const float x = (float)a.row[ray], y = (float)a.col[ray];
  float p[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) p[j] = (x * a.K[j] + y * a.K[3 + j]) + a.K[6 + j];
  // F.normalize (nerf.py:193): ATen's CPU 2-norm accumulates acc = fma(v, v, acc) in fp32 and takes the
  // square root in double; clamp_min(1e-12); true division.
  const float ss = __builtin_fmaf(p[2], p[2], __builtin_fmaf(p[1], p[1], p[0] * p[0]));
  float nrm = (float)sqrt((double)ss);
  nrm = fmaxf(nrm, 1e-12f);
  float d[3], dw[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) d[j] = p[j] / nrm;
#pragma unroll
  for (int c = 0; c < 3; ++c) dw[c] = (pb[5 * c] * d[0] + pb[5 * c + 1] * d[1]) + pb[5 * c + 2] * d[2];
  const float near = pb[15], far = pb[16];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[RF_R + 3 * c + k] = pb[5 * c + k];
    rec[RF_O + c] = pb[5 * c + 3];
    rec[RF_DCAM + c] = d[c];
    rec[RF_DWRD + c] = dw[c];
  }
  rec[RF_NEAR] = near;
  rec[RF_FAR] = far;
  rec[RF_STEP] = (far - near) / (float)(a.Nc - 1);
  rec[RF_DELTA] = (far - near) / (float)a.Nc;  // quirk Q5 (nerf.py:293)
  rec[22] = 0.f;
  rec[23] = 0.f;
}
// coarse depth i of a ray (nerf.py:288: numpy.linspace(near, far, Nc) in fp32 -- separate multiply and add, the end point exact)
__device__ __forceinline__ float coarse_depth(float near, float far, float step, int i, int Nc) { return (i == Nc - 1) ? far : ((float)i * step + near); }

// one ray by 128 consecutive threads (tid = 0..127 inside the group); the dvec part ends in a __syncthreads(): with a.dvec every thread
// of the block must call this
__device__ __forceinline__ void ray_block(const RaysArgs& a, const int ray, const int tid, float* gd /* LDS, DIR_DIM floats per group */) {
  if (a.status && ray == 0 && tid < STATUS_STICKY_WORD) a.status[tid] = 0u;  // the forward's status words start clean (instead of a memset node of their own); the sticky ones stay
  float rec[RAYF];
  ray_record(a, ray, rec);
  const float near = rec[RF_NEAR], far = rec[RF_FAR], step = rec[RF_STEP];
  if (tid == 0) {
    if (a.rayf) {
      float* rf = a.rayf + (size_t)ray * RAYF;
#pragma unroll
      for (int k = 0; k < RAYF; ++k) rf[k] = rec[k];
    }
    if (a.d_cam)
      for (int c = 0; c < 3; ++c) a.d_cam[(size_t)ray * 3 + c] = rec[RF_DCAM + c];
    if (a.d_wrd)
      for (int c = 0; c < 3; ++c) a.d_wrd[(size_t)ray * 3 + c] = rec[RF_DWRD + c];
  }
  if (a.t_c) {
    for (int i = tid; i < a.Nc; i += 128) a.t_c[(size_t)ray * a.Nc + i] = coarse_depth(near, far, step, i, a.Nc);
  }
  if (a.dvec) {
    if (tid < 12) {
      const int c = tid >> 2, l = tid & 3;
      const float ph = rec[RF_DWRD + c] * __uint_as_float(kFreqDirBits[l]);
      gd[c * 8 + 2 * l] = sinf(ph);
      gd[c * 8 + 2 * l + 1] = cosf(ph);
    }
    __syncthreads();
    const float* wr = a.w_dir + (size_t)tid * (WIDTH + DIR_DIM);
    float s = a.b_dir[tid];
#pragma unroll
    for (int k = 0; k < DIR_DIM; ++k) s = __builtin_fmaf(wr[k], gd[k], s);
    if (a.b_fold) s += a.b_fold[tid];  // point_info's bias through dir_info's feature columns (common.h SEG_FOLD)
    a.dvec[(size_t)ray * HALF + tid] = s;
  }
}

}  // namespace nerf
