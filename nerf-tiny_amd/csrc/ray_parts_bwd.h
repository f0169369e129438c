// ray_parts_bwd.h -- backward of the per-ray stages as device functions of ONE 64-lane wave (k_merge_bwd / k_coarse_bwd of ray_ops_bwd.hip are
// made of them; the bf16 chain kernels call them as prologues for small batches: field_bwd_bf16.hip).  No workgroup barrier inside: the
// caller passes the `sync` that fits its workgroup.  MI355X / gfx950 only; built with -ffp-contract=off.
#pragma once
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

// backward of the merged composite + channel sorts for ONE ray by one wave (nerf.py:302-321); sm_f: 4 * (Nc + Nf) floats of LDS of this wave;
// `sync` orders this wave's LDS writes before its reads (a one-wave workgroup passes __syncthreads, a wave of a bigger workgroup a wave fence)
template <class Sync>
__device__ __forceinline__ void merge_bwd_ray(const MergeBwdArgs& a, const int ray, const int lane, float* sm_f, Sync&& sync) {
  const int N = a.Nc + a.Nf;
  float* s_te = sm_f;          // T_k * exp(-s_k)
  float* s_dww = sm_f + N;     // dw_k * w_k
  float* s_dw = sm_f + 2 * N;  // dw_k
  float* s_dd = sm_f + 3 * N;  // d delta_k
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
  sync();
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
  sync();
  // pass 3: delta_k = t_{k+1} - t_k (k < N-1), delta_{N-1} constant  =>  d t_k = d delta_{k-1} - d delta_k
  for (int i = lane; i < N; i += 64) {
    float g = 0.f;
    if (i > 0) g += s_dd[i - 1];
    if (i + 1 < N) g -= s_dd[i];
    const int src = pm[i];
    if (src >= a.Nc) a.dt_f[(size_t)ray * a.Nf + (src - a.Nc)] = g;  // t_coarse carries no gradient
  }
}


// backward of resampling + coarse composite for ONE ray by one wave (nerf.py:225-281); w: 5 * Nc floats, ks: Nf u16 of LDS of this wave
template <class Sync>
__device__ __forceinline__ void coarse_bwd_ray(const CoarseBwdArgs& a, const int ray, const bool live, const int lane, float* w, uint16_t* ks, Sync&& sync) {
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
  float* cdf = w + Nc;
  float* xs = w + 2 * Nc;   // T * e
  float* ys = w + 3 * Nc;   // dw contribution to bin + 1
  float* dcs = w + 4 * Nc;  // d cdf

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
  sync();
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
  sync();
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
  sync();
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


}  // namespace nerf
