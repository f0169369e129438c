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

namespace nerf {

__device__ __forceinline__ double wave_incl_scan_d(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
// inclusive suffix sum: result(lane) = sum_{l >= lane} v(l)
__device__ __forceinline__ double wave_suffix_scan_d(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_down(v, d);
    if (lane + d < 64) v += o;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_merge_bwd: grid = B blocks of 64 threads; dynamic LDS = 4 * N floats
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_merge_bwd(const MergeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  const int N = a.Nc + a.Nf;
  float* s_te = sm_f;          // T_k * exp(-s_k)
  float* s_dww = sm_f + N;     // dw_k * w_k
  float* s_dw = sm_f + 2 * N;  // dw_k
  float* s_dd = sm_f + 3 * N;  // d delta_k
  const int lane = threadIdx.x, ray = blockIdx.x;
  const float* bun = a.bundle + (size_t)ray * N * 5;
  const float dC0 = a.dC_f[(size_t)ray * 3], dC1 = a.dC_f[(size_t)ray * 3 + 1], dC2 = a.dC_f[(size_t)ray * 3 + 2];
  const uint16_t* pm = a.perm + (size_t)ray * 5 * N;

  // pass 1 (forward): recompute T, w exactly as k_merge did; d rgb_sorted scattered immediately
  double carry = 0.0;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    const bool v = i < N;
    const float ti = v ? bun[(size_t)i * 5] : 0.f;
    const float dl = (v && i + 1 < N) ? (bun[(size_t)(i + 1) * 5] - ti) : a.last;
    const float sg = v ? bun[(size_t)i * 5 + 4] : 0.f;
    const float s = v ? dl * sg : 0.f;
    double cs = wave_incl_scan_d((double)s, lane) + carry;
    carry = __shfl(cs, 63);
    const float T = expf(-(float)cs);
    const float e = expf(-s);
    const float wi = T * (1.0f - e);
    if (v) {
      const float r = bun[(size_t)i * 5 + 1], g = bun[(size_t)i * 5 + 2], b = bun[(size_t)i * 5 + 3];
      const float dw = __builtin_fmaf(b, dC2, __builtin_fmaf(g, dC1, r * dC0));
      s_te[i] = T * e;
      s_dw[i] = dw;
      s_dww[i] = dw * wi;
      // d rgb (sorted position i, channel c) = w_i * dC_c -> original sample perm[1+c][i]
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int src = pm[(size_t)(1 + c) * N + i];
        const float gch = wi * (c == 0 ? dC0 : (c == 1 ? dC1 : dC2));
        if (src < a.Nc) a.drgb_c[((size_t)ray * a.Nc + src) * 3 + c] = gch;
        else a.drgb_f[((size_t)ray * a.Nf + (src - a.Nc)) * 3 + c] = gch;
      }
    }
  }
  __syncthreads();
  // pass 2 (reverse): ds_k = dw_k T_k e_k - sum_{i>=k} dw_i w_i ; d sigma_k = ds_k delta_k ; d delta_k = ds_k sigma_k
  double rc = 0.0;
  const int nchunks = (N + 63) / 64;
  for (int ch = nchunks - 1; ch >= 0; --ch) {
    const int i = ch * 64 + lane;
    const bool v = i < N;
    const double x = v ? (double)s_dww[i] : 0.0;
    const double suf = wave_suffix_scan_d(x, lane) + rc;
    rc = __shfl(suf, 0);
    if (v) {
      const float ds = s_dw[i] * s_te[i] - (float)suf;
      const float ti = bun[(size_t)i * 5];
      const float dl = (i + 1 < N) ? (bun[(size_t)(i + 1) * 5] - ti) : a.last;
      const float sg = bun[(size_t)i * 5 + 4];
      s_dd[i] = ds * sg;
      const float dsg = ds * dl;
      const int src = pm[(size_t)4 * N + i];
      if (src < a.Nc) a.dsig_c[(size_t)ray * a.Nc + src] = dsg;
      else a.dsig_f[(size_t)ray * a.Nf + (src - a.Nc)] = dsg;
    }
  }
  __syncthreads();
  // pass 3: delta_k = t_{k+1} - t_k (k < N-1), delta_{N-1} constant  =>  d t_k = d delta_{k-1} - d delta_k
  for (int i = lane; i < N; i += 64) {
    float g = 0.f;
    if (i > 0) g += s_dd[i - 1];
    if (i + 1 < N) g -= s_dd[i];
    const int src = pm[i];
    if (src >= a.Nc) a.dt_f[(size_t)ray * a.Nf + (src - a.Nc)] = g;  // t_coarse carries no gradient
  }
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
  const int Nc = a.Nc, Nf = a.Nf;
  float near, far;
  if (a.rayf) {
    near = a.rayf[(size_t)ray * RAYF + RF_NEAR];
    far = a.rayf[(size_t)ray * RAYF + RF_FAR];
  } else {
    near = a.near_far[2 * ray];
    far = a.near_far[2 * ray + 1];
  }
  const float delta_c = (far - near) / (float)Nc;
  float delta0 = a.delta0;
  if (a.delta0_mode == 0) {
    const float n0 = a.ray0_override ? a.near0 : a.rayf[RF_NEAR];
    const float f0 = a.ray0_override ? a.far0 : a.rayf[RF_FAR];
    const float st0 = (f0 - n0) / (float)(Nc - 1);
    const float t1 = (Nc == 2) ? f0 : (1.0f * st0 + n0);
    delta0 = t1 - n0;
  }
  float* w = sm_c + (size_t)wv * 5 * Nc;
  float* cdf = w + Nc;
  float* xs = w + 2 * Nc;   // T * e
  float* ys = w + 3 * Nc;   // dw contribution to bin + 1
  float* dcs = w + 4 * Nc;  // d cdf
  uint16_t* ks = reinterpret_cast<uint16_t*>(sm_c + (size_t)4 * 5 * Nc) + (size_t)wv * ((Nf + 1) & ~1);

  // forward recompute (identical arithmetic to k_coarse)
  double carry = 0.0, carry2 = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (int base = 0; base < Nc; base += 64) {
    const int i = base + lane;
    const bool v = i < Nc;
    const size_t gi = (size_t)ray * Nc + (v ? i : 0);
    const float sg = v ? a.sigma[gi] : 0.f;
    const float s = delta_c * sg;
    double cs = wave_incl_scan_d((double)s, lane) + carry;
    carry = __shfl(cs, 63);
    const float T = expf(-(float)cs);
    const float e = expf(-s);
    const float wi = v ? T * (1.0f - e) : 0.f;
    double cw = wave_incl_scan_d((double)wi, lane) + carry2;
    carry2 = __shfl(cw, 63);
    const float cd = (float)cw;
    if (v) {
      w[i] = wi;
      cdf[i] = cd;
      xs[i] = T * e;
      lo = fminf(lo, cd);
      hi = fmaxf(hi, cd);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d));
    hi = fmaxf(hi, __shfl_xor(hi, d));
  }
  __syncthreads();
  const float step = (hi - lo) / (float)(Nf + 1);
  for (int j = lane; j < Nf; j += 64) {
    const float u = (float)(j + 1) * step + lo;
    int lo_i = 0, hi_i = Nc;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (cdf[mid] < u) lo_i = mid + 1; else hi_i = mid;
    }
    int k = lo_i - 1;
    k = k < 0 ? 0 : (k > Nc - 1 ? Nc - 1 : k);
    ks[j] = (uint16_t)k;
  }
  __syncthreads();
  // per coarse bin i: gather the fine samples that fell into it (k_j is non-decreasing in j)
  for (int base = 0; base < Nc; base += 64) {
    const int i = base + lane;
    float dcdf = 0.f, dwn = 0.f;
    if (i < Nc) {
      int b0 = 0, b1 = Nf;
      while (b0 < b1) { const int mid = (b0 + b1) >> 1; if (ks[mid] < i) b0 = mid + 1; else b1 = mid; }
      const int jb = b0;
      b1 = Nf;
      while (b0 < b1) { const int mid = (b0 + b1) >> 1; if (ks[mid] <= i) b0 = mid + 1; else b1 = mid; }
      const int je = b0;
      if (i + 1 < Nc && je > jb) {
        const float den = w[i + 1] + 1e-7f;
        const float slope = delta0 / den;
        const float dslope = -(delta0 / (den * den));  // d slope / d w[i+1]
        const float ci = cdf[i];
        for (int j = jb; j < je; ++j) {
          const float u = (float)(j + 1) * step + lo;
          const float g = a.dt_f[(size_t)ray * Nf + j];
          dcdf -= g * slope;
          dwn += g * (u - ci) * dslope;
        }
      }
    }
    if (i < Nc) {
      dcs[i] = dcdf;
      ys[i] = dwn;
    }
  }
  __syncthreads();
  const float dC0 = a.dC_c[(size_t)ray * 3], dC1 = a.dC_c[(size_t)ray * 3 + 1], dC2 = a.dC_c[(size_t)ray * 3 + 2];
  // dw_i = suffix_sum(dcdf)_i + dwn_{i-1} + rgb_i . dC_c ; then composite backward
  double rc = 0.0, rc2 = 0.0;
  const int nchunks = (Nc + 63) / 64;
  for (int ch = nchunks - 1; ch >= 0; --ch) {
    const int i = ch * 64 + lane;
    const bool v = i < Nc;
    const double sufc = wave_suffix_scan_d(v ? (double)dcs[i] : 0.0, lane) + rc;
    rc = __shfl(sufc, 0);
    const size_t gi = (size_t)ray * Nc + (v ? i : 0);
    float dw = 0.f;
    if (v) {
      dw = (float)sufc + (i > 0 ? ys[i - 1] : 0.f);
      dw += __builtin_fmaf(a.rgb[gi * 3 + 2], dC2, __builtin_fmaf(a.rgb[gi * 3 + 1], dC1, a.rgb[gi * 3] * dC0));
    }
    const double suf = wave_suffix_scan_d(v ? (double)(dw * w[i]) : 0.0, lane) + rc2;
    rc2 = __shfl(suf, 0);
    if (v && live) {
      const float ds = dw * xs[i] - (float)suf;
      a.dsig_c[gi] += ds * delta_c;
      a.drgb_c[gi * 3 + 0] += w[i] * dC0;
      a.drgb_c[gi * 3 + 1] += w[i] * dC1;
      a.drgb_c[gi * 3 + 2] += w[i] * dC2;
    }
  }
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
