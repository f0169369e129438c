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

namespace nerf {

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}

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
constexpr int MAXN = 1024;


__global__ __launch_bounds__(256) void k_coarse(const CoarseArgs a) {
  __shared__ float s_w[4][MAXN];
  __shared__ float s_cdf[4][MAXN];
  __shared__ float s_t[4][MAXN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * 4 + wv;
  const bool live = ray_raw < a.B;
  const int ray = live ? ray_raw : a.B - 1;
  float near, far;
  if (a.rayf) {
    near = a.rayf[(size_t)ray * RAYF + RF_NEAR];
    far = a.rayf[(size_t)ray * RAYF + RF_FAR];
  } else {
    near = a.near_far[2 * ray];
    far = a.near_far[2 * ray + 1];
  }
  const float delta_c = (far - near) / (float)a.Nc;
  // spacing of RAY 0 used for every ray (quirk Q6, nerf.py:233): t[0][1] - t[0][0]
  float delta0 = a.delta0;
  if (a.delta0_mode == 0) {
    const float n0 = a.ray0_override ? a.near0 : a.rayf[RF_NEAR];
    const float f0 = a.ray0_override ? a.far0 : a.rayf[RF_FAR];
    const float st0 = (f0 - n0) / (float)(a.Nc - 1);
    const float t1 = (a.Nc == 2) ? f0 : (1.0f * st0 + n0);
    delta0 = t1 - n0;
  }
  float* w = s_w[wv];
  float* cdf = s_cdf[wv];
  float* tc = s_t[wv];

  double carry = 0.0, carry2 = 0.0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  float lo = INFINITY, hi = -INFINITY;
  for (int base = 0; base < a.Nc; base += 64) {
    const int i = base + lane;
    const bool v = i < a.Nc;
    const size_t gi = (size_t)ray * a.Nc + (v ? i : 0);
    const float sg = v ? a.sigma[gi] : 0.f;
    const float s = delta_c * sg;
    double cs = wave_incl_scan((double)s, lane) + carry;
    carry = __shfl(cs, 63);
    const float T = expf(-(float)cs);
    const float wi = v ? T * (1.0f - expf(-s)) : 0.f;
    double cw = wave_incl_scan((double)wi, lane) + carry2;
    carry2 = __shfl(cw, 63);
    const float cd = (float)cw;
    if (v) {
      w[i] = wi;
      cdf[i] = cd;
      tc[i] = a.t_c[gi];
      if (live && a.w_c) a.w_c[gi] = wi;
      c0 += wi * a.rgb[gi * 3 + 0];
      c1 += wi * a.rgb[gi * 3 + 1];
      c2 += wi * a.rgb[gi * 3 + 2];
      lo = fminf(lo, cd);
      hi = fmaxf(hi, cd);
    }
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  lo = wave_min(lo);
  hi = wave_max(hi);
  if (live && lane == 0 && a.C_coarse) {
    a.C_coarse[(size_t)ray * 3 + 0] = c0;
    a.C_coarse[(size_t)ray * 3 + 1] = c1;
    a.C_coarse[(size_t)ray * 3 + 2] = c2;
  }
  __syncthreads();
  // u_j = lo + j * ((hi - lo)/(Nf+1)), j = 1..Nf  (numpy.linspace(lo, hi, Nf+2)[1:-1] in fp32, nerf.py:243-246)
  const float step = (hi - lo) / (float)(a.Nf + 1);
  bool bad = false;
  for (int j = lane; j < a.Nf; j += 64) {
    const float u = (float)(j + 1) * step + lo;
    // searchsorted(cdf, u) (left) = number of cdf entries < u
    int lo_i = 0, hi_i = a.Nc;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (cdf[mid] < u) lo_i = mid + 1; else hi_i = mid;
    }
    int k = lo_i - 1;
    if (k > a.Nf - 1 || k < 0) bad = true;  // the condition of nerf.py:251 (quirk Q7)
    k = k < 0 ? 0 : (k > a.Nc - 1 ? a.Nc - 1 : k);
    const float slope = (k + 1 < a.Nc) ? delta0 / (w[k + 1] + 1e-7f) : 0.f;
    const float tf = tc[k] + (u - cdf[k]) * slope;
    if (live) a.t_f[(size_t)ray * a.Nf + j] = tf;
  }
  if (live && bad && a.status) atomicOr(a.status, 1u);
  if (live && bad && a.sticky) atomicOr(a.sticky, 1u);
}

// ---------------------------------------------------------------------------------------------
// k_merge: nerf.py:302-321.  cat -> sort(dim=1) of a [B,N,5] bundle = five INDEPENDENT ascending channel
// sorts (quirk Q1), delta_i = t_{i+1} - t_i with the last = `last`, weights, C_fine.
// One wave (64-thread block) per ray; bitonic sort over P = next_pow2(N) slots per channel in LDS.
// Ties are broken by original index (a stable sort); indices are kept for the backward pass.
// ---------------------------------------------------------------------------------------------

// Bitonic sort of five independent 256-slot channels by one wave IN REGISTERS: lane l holds slots 4l .. 4l+3 of every channel, so
// the 15 stages with partner distance 1 or 2 are lane-local and the 21 others exchange with lane l ^ (distance / 4) through DPP
// (distance 4, 8), ds_swizzle (16 .. 64) or one permute (128) -- no LDS traffic between stages and no barriers (the LDS version
// spent 36 barriers and ~500 two-address LDS instructions per ray).  Keys are the floats' order-preserving unsigned images
// (sign flipped for positives, all bits for negatives; -0 keyed as +0 and NaN as the maximum: the order of torch.sort); WITH_IDX carries
// the original slot as the low half of a 64-bit key, which makes the order total (= a stable sort, as before).
template <int D>
__device__ __forceinline__ unsigned lane_xor_get(unsigned v) {  // v of lane (l ^ D)
  if constexpr (D == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  else if constexpr (D < 32) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (D << 10));   // bit-mask mode: xor D inside 32 lanes
  else return (unsigned)__shfl_xor((int)v, 32);
}
// torch.sort's order: -0 and +0 compare equal (both map to +0's key; the original index breaks the tie, as everywhere), every NaN
// sorts last (the maximum key, above the +inf padding).
__device__ __forceinline__ unsigned sort_key(float x) {
  unsigned b = __float_as_uint(x);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;  // NaN of either sign
  if (b == 0x80000000u) b = 0u;                              // -0 -> +0
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float sort_unkey(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

template <bool WITH_IDX, int K, int J>
__device__ __forceinline__ void sortreg_stage(unsigned (&key)[5][4], unsigned (&ix)[5][4], int lane) {
  if constexpr (J >= 4) {
    constexpr int D = J / 4;
    const bool upper = (lane & D) != 0;
    const bool asc = K >= 256 ? true : ((4 * lane) & K) == 0;
    const bool flip = upper != !asc;  // take the partner's element iff (mine > partner's) != flip
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned pk = lane_xor_get<D>(key[c][r]);
        bool gt;
        unsigned pi = 0;
        if (WITH_IDX) {
          pi = lane_xor_get<D>(ix[c][r]);
          gt = (((unsigned long long)key[c][r] << 32) | ix[c][r]) > (((unsigned long long)pk << 32) | pi);
        } else {
          gt = key[c][r] > pk;
        }
        const bool take = gt != flip;
        key[c][r] = take ? pk : key[c][r];
        if (WITH_IDX) ix[c][r] = take ? pi : ix[c][r];
      }
  } else {
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r & J) continue;
        const int q = r | J;
        const bool asc = K >= 256 ? true : ((4 * lane + r) & K) == 0;
        const unsigned a = key[c][r], bb = key[c][q];
        bool gt;
        if (WITH_IDX) gt = (((unsigned long long)a << 32) | ix[c][r]) > (((unsigned long long)bb << 32) | ix[c][q]);
        else gt = a > bb;
        const bool sw = gt == asc;
        key[c][r] = sw ? bb : a;
        key[c][q] = sw ? a : bb;
        if (WITH_IDX) {
          const unsigned ia = ix[c][r], ib = ix[c][q];
          ix[c][r] = sw ? ib : ia;
          ix[c][q] = sw ? ia : ib;
        }
      }
  }
}
template <bool WITH_IDX, int K, int J>
__device__ __forceinline__ void sortreg_merge(unsigned (&key)[5][4], unsigned (&ix)[5][4], int lane) {
  sortreg_stage<WITH_IDX, K, J>(key, ix, lane);
  if constexpr (J > 1) sortreg_merge<WITH_IDX, K, J / 2>(key, ix, lane);
}
template <bool WITH_IDX, int K>
__device__ __forceinline__ void sortreg_from(unsigned (&key)[5][4], unsigned (&ix)[5][4], int lane) {
  sortreg_merge<WITH_IDX, K, K / 2>(key, ix, lane);
  if constexpr (K < 256) sortreg_from<WITH_IDX, K * 2>(key, ix, lane);
}
// val [5][256] floats (and idx [5][256] u16, WITH_IDX) in LDS: read as 4 slots per lane, sort, write back in sorted order
template <bool WITH_IDX>
__device__ __forceinline__ void sort256_regs(float* val, uint16_t* idx, int lane) {
  unsigned key[5][4], ix[5][4];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(val + c * 256 + 4 * lane);
    key[c][0] = sort_key(v.x); key[c][1] = sort_key(v.y); key[c][2] = sort_key(v.z); key[c][3] = sort_key(v.w);
#pragma unroll
    for (int r = 0; r < 4; ++r) ix[c][r] = (unsigned)(4 * lane + r);  // (= what the loader wrote to idx)
  }
  sortreg_from<WITH_IDX, 2>(key, ix, lane);
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    *reinterpret_cast<float4*>(val + c * 256 + 4 * lane) =
        make_float4(sort_unkey(key[c][0]), sort_unkey(key[c][1]), sort_unkey(key[c][2]), sort_unkey(key[c][3]));
    if (WITH_IDX) {
      uint2 pk;
      pk.x = ix[c][0] | (ix[c][1] << 16);
      pk.y = ix[c][2] | (ix[c][3] << 16);
      *reinterpret_cast<uint2*>(idx + c * 256 + 4 * lane) = pk;
    }
  }
  __syncthreads();
}

template <bool WITH_IDX>  // WITH_IDX: carry the original index (stable order + permutation for backward)
__global__ __launch_bounds__(64) void k_merge(const MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int P = a.P, N = a.Nc + a.Nf;
  float* val = reinterpret_cast<float*>(smem_raw);                  // [5][P]
  uint16_t* idx = reinterpret_cast<uint16_t*>(val + 5 * (size_t)P);  // [5][P]
  const int lane = threadIdx.x;
  const int ray = blockIdx.x;
  // load: channel 0 = t, 1..3 = rgb, 4 = sigma
  for (int i = lane; i < P; i += 64) {
    float v[5];
    if (i < a.Nc) {
      const size_t g = (size_t)ray * a.Nc + i;
      v[0] = a.t_c[g]; v[1] = a.rgb_c[g * 3]; v[2] = a.rgb_c[g * 3 + 1]; v[3] = a.rgb_c[g * 3 + 2]; v[4] = a.sig_c[g];
    } else if (i < N) {
      const size_t g = (size_t)ray * a.Nf + (i - a.Nc);
      v[0] = a.t_f[g]; v[1] = a.rgb_f[g * 3]; v[2] = a.rgb_f[g * 3 + 1]; v[3] = a.rgb_f[g * 3 + 2]; v[4] = a.sig_f[g];
    } else {
      v[0] = v[1] = v[2] = v[3] = v[4] = __builtin_nanf("");  // padding = the maximum key: behind every real value, NaNs included
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      val[c * P + i] = v[c];
      if (WITH_IDX) idx[c * P + i] = (uint16_t)i;
    }
  }
  __syncthreads();
  if (P == 256) {  // the usual size (64 + 128 samples): the whole network in registers
    sort256_regs<WITH_IDX>(val, idx, lane);
  } else
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int tI = lane; tI < (P >> 1); tI += 64) {
        const int i = ((tI & ~(j - 1)) << 1) | (tI & (j - 1));  // element with bit j clear
        const int l = i | j;
        const bool asc = (i & k) == 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const float x = val[c * P + i], y = val[c * P + l];
          if (WITH_IDX) {
            const uint16_t xi = idx[c * P + i], yi = idx[c * P + l];
            const unsigned kx = sort_key(x), ky = sort_key(y);  // the register network's order: +-0 equal, NaN last
            const bool gt = (kx > ky) || (kx == ky && xi > yi);
            if (gt == asc) {
              val[c * P + i] = y; val[c * P + l] = x;
              idx[c * P + i] = yi; idx[c * P + l] = xi;
            }
          } else {  // values only: equal keys are interchangeable
            const unsigned kx = sort_key(x), ky = sort_key(y);
            const bool gt = kx > ky;
            if (gt == asc && kx != ky) {
              val[c * P + i] = y; val[c * P + l] = x;
            }
          }
        }
      }
      __syncthreads();
    }
  }
  // composite over the sorted channels
  double carry = 0.0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    const bool v = i < N;
    const float ti = v ? val[i] : 0.f;
    const float dl = (v && i + 1 < N) ? (val[i + 1] - ti) : a.last;
    const float sg = v ? val[4 * P + i] : 0.f;
    const float s = v ? dl * sg : 0.f;
    double cs = wave_incl_scan((double)s, lane) + carry;
    carry = __shfl(cs, 63);
    const float T = expf(-(float)cs);
    const float wi = v ? T * (1.0f - expf(-s)) : 0.f;
    if (v) {
      const float r = val[P + i], g = val[2 * P + i], b = val[3 * P + i];
      c0 += wi * r; c1 += wi * g; c2 += wi * b;
      const size_t gi = (size_t)ray * N + i;
      if (a.w) a.w[gi] = wi;
      if (a.bundle) {
        float* o = a.bundle + gi * 5;
        o[0] = ti; o[1] = r; o[2] = g; o[3] = b; o[4] = sg;
      }
      if (WITH_IDX && a.perm) {
#pragma unroll
        for (int c = 0; c < 5; ++c) a.perm[((size_t)ray * 5 + c) * N + i] = idx[c * P + i];
      }
    }
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  if (lane == 0) {
    a.C_fine[(size_t)ray * 3 + 0] = c0;
    a.C_fine[(size_t)ray * 3 + 1] = c1;
    a.C_fine[(size_t)ray * 3 + 2] = c2;
  }
}

// ---------------------------------------------------------------------------------------------
// k_ray_loss: nerf.py:325-331 (sum, not mean) and d loss / dC.  Single 1024-thread block (B*3 elements).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ray_loss(const float* Cc, const float* Cf, const float* Ct, int n, float* loss, float* dCc,
                                                   float* dCf) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float e1 = Cc[i] - Ct[i], e2 = Cf[i] - Ct[i];
    acc += e1 * e1 + e2 * e2;
    if (dCc) dCc[i] = 2.0f * e1;
    if (dCf) dCf[i] = 2.0f * e2;
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
  if (a.perm) {
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
