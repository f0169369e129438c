// ray_parts.h -- the per-ray stages of the hot path as device functions of ONE 64-lane wave: coarse weights + C_coarse, inverse-CDF
// resampling, the five channel sorts and the merged composite.  Used by their stand-alone kernels (k_coarse, k_merge: ray_ops.hip) and by the
// fused small-batch kernel that renders a pair of rays in one workgroup (field_pair_bf16x.hip) -- one source, the same bits.
// None of these functions contains a workgroup barrier: callers whose workgroup is one wave per ray pass __syncthreads() as `sync`, callers
// that run them on SOME waves of a bigger workgroup pass a wave-level fence.  MI355X / gfx950 only; built with -ffp-contract=off.
#pragma once
#include "kernels.h"

namespace nerf {

constexpr int MAXN = 1024;

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


// coarse spacing of ray 0 as the reference takes it (quirk Q6, nerf.py:233): t[0][1] - t[0][0] of numpy.linspace(near, far, Nc)
__device__ __forceinline__ float ray0_spacing(float n0, float f0, int Nc) {
  const float st0 = (f0 - n0) / (float)(Nc - 1);
  const float t1 = (Nc == 2) ? f0 : (1.0f * st0 + n0);
  return t1 - n0;
}

// get_density (nerf.py:263-272) with delta = (far - near) / Nc (quirk Q5), color_cum (nerf.py:274-281): the ray's Nc coarse samples at
// sigma[i], rgb[3 i + ch], t_c[i] (global or LDS) -> w, cdf, tc (LDS, this wave's), optionally w_c (global) and C_coarse[3]; lo / hi =
// min / max of the cdf (nerf.py:240-241).  The caller orders the LDS writes before coarse_ray_resample's reads.
__device__ __forceinline__ void coarse_ray_weights(const float* sigma, const float* rgb, const float* t_c, float near, float far, int Nc, int lane,
                                                   float* w, float* cdf, float* tc, float* w_c_out, float* C_out, float& lo_out, float& hi_out) {
  const float delta_c = (far - near) / (float)Nc;
  double carry = 0.0, carry2 = 0.0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  float lo = INFINITY, hi = -INFINITY;
  for (int base = 0; base < Nc; base += 64) {
    const int i = base + lane;
    const bool v = i < Nc;
    const int gi = v ? i : 0;
    const float sg = v ? sigma[gi] : 0.f;
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
      tc[i] = t_c[gi];
      if (w_c_out) w_c_out[gi] = wi;
      c0 += wi * rgb[gi * 3 + 0];
      c1 += wi * rgb[gi * 3 + 1];
      c2 += wi * rgb[gi * 3 + 2];
      lo = fminf(lo, cd);
      hi = fmaxf(hi, cd);
    }
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  lo_out = wave_min(lo);
  hi_out = wave_max(hi);
  if (lane == 0 && C_out) {
    C_out[0] = c0;
    C_out[1] = c1;
    C_out[2] = c2;
  }
}

// resample (nerf.py:225-261): u_j = lo + j * ((hi - lo)/(Nf+1)), j = 1..Nf (numpy.linspace(lo, hi, Nf+2)[1:-1] in fp32, nerf.py:243-246),
// searchsorted (left) in the cdf, t_f = t_c[k] + (u - cdf[k]) * delta0 / (w[k+1] + 1e-7).  Returns whether any lane met the condition of
// nerf.py:251 (quirk Q7: the index is clamped here, the caller reports).  t_f_out: Nf floats (global or LDS) or null.
__device__ __forceinline__ bool coarse_ray_resample(const float* w, const float* cdf, const float* tc, float lo, float hi, float delta0, int Nc, int Nf,
                                                    int lane, float* t_f_out) {
  const float step = (hi - lo) / (float)(Nf + 1);
  bool bad = false;
  for (int j = lane; j < Nf; j += 64) {
    const float u = (float)(j + 1) * step + lo;
    // searchsorted(cdf, u) (left) = number of cdf entries < u
    int lo_i = 0, hi_i = Nc;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (cdf[mid] < u) lo_i = mid + 1; else hi_i = mid;
    }
    int k = lo_i - 1;
    if (k > Nf - 1 || k < 0) bad = true;  // the condition of nerf.py:251 (quirk Q7)
    k = k < 0 ? 0 : (k > Nc - 1 ? Nc - 1 : k);
    const float slope = (k + 1 < Nc) ? delta0 / (w[k + 1] + 1e-7f) : 0.f;
    const float tf = tc[k] + (u - cdf[k]) * slope;
    if (t_f_out) t_f_out[j] = tf;
  }
  return bad;
}

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

template <bool WITH_IDX, int K, int J, int NCH>
__device__ __forceinline__ void sortreg_stage(unsigned (&key)[NCH][4], unsigned (&ix)[NCH][4], int lane) {
  if constexpr (J >= 4) {
    constexpr int D = J / 4;
    const bool upper = (lane & D) != 0;
    const bool asc = K >= 256 ? true : ((4 * lane) & K) == 0;
    const bool flip = upper != !asc;  // take the partner's element iff (mine > partner's) != flip
#pragma unroll
    for (int c = 0; c < NCH; ++c)
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
    for (int c = 0; c < NCH; ++c)
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
template <bool WITH_IDX, int K, int J, int NCH>
__device__ __forceinline__ void sortreg_merge(unsigned (&key)[NCH][4], unsigned (&ix)[NCH][4], int lane) {
  sortreg_stage<WITH_IDX, K, J, NCH>(key, ix, lane);
  if constexpr (J > 1) sortreg_merge<WITH_IDX, K, J / 2, NCH>(key, ix, lane);
}
template <bool WITH_IDX, int K, int NCH>
__device__ __forceinline__ void sortreg_from(unsigned (&key)[NCH][4], unsigned (&ix)[NCH][4], int lane) {
  sortreg_merge<WITH_IDX, K, K / 2, NCH>(key, ix, lane);
  if constexpr (K < 256) sortreg_from<WITH_IDX, K * 2, NCH>(key, ix, lane);
}
// val [5][256] floats (and idx [5][256] u16, WITH_IDX) in LDS: read as 4 slots per lane, sort, write back in sorted order
template <bool WITH_IDX, class Sync>
__device__ __forceinline__ void sort256_regs(float* val, uint16_t* idx, int lane, Sync&& sync) {
  unsigned key[5][4], ix[5][4];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(val + c * 256 + 4 * lane);
    key[c][0] = sort_key(v.x); key[c][1] = sort_key(v.y); key[c][2] = sort_key(v.z); key[c][3] = sort_key(v.w);
#pragma unroll
    for (int r = 0; r < 4; ++r) ix[c][r] = (unsigned)(4 * lane + r);  // (= what the loader wrote to idx)
  }
  sortreg_from<WITH_IDX, 2, 5>(key, ix, lane);
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
  sync();
}


// ONE channel of 256 slots (values only): lane l holds slots 4l .. 4l+3 in v; the same network, the same compare-exchanges as a channel of
// sort256_regs -- so a channel sorted alone (the ray-pair kernel deals a pair's ten channel sorts out over its eight waves) comes out
// bit for bit as inside the five-channel sort.
__device__ __forceinline__ void sort256_one_channel(float (&v)[4], int lane) {
  unsigned key[1][4], ix[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { key[0][r] = sort_key(v[r]); ix[0][r] = 0u; }
  sortreg_from<false, 2, 1>(key, ix, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = sort_unkey(key[0][r]);
}
// ... and WITH the original slots (the training form: a stable sort whose permutation the backward un-sorts through): on return ix[r] =
// the slot the value at sorted position 4 lane + r came from
__device__ __forceinline__ void sort256_one_channel_idx(float (&v)[4], unsigned (&ixo)[4], int lane) {
  unsigned key[1][4], ix[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { key[0][r] = sort_key(v[r]); ix[0][r] = (unsigned)(4 * lane + r); }
  sortreg_from<true, 2, 1>(key, ix, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) { v[r] = sort_unkey(key[0][r]); ixo[r] = ix[0][r]; }
}
// channel `ch` (0 = t, 1..3 = rgb, 4 = sigma) of slot i of a ray's merged bundle as k_merge loads it: coarse samples first, then fine, NaN padding
__device__ __forceinline__ float merge_slot_value(const MergeArgs& a, const int ray, const int ch, const int i) {
  const int N = a.Nc + a.Nf;
  if (i < a.Nc) {
    const size_t g = (size_t)ray * a.Nc + i;
    return ch == 0 ? a.t_c[g] : ch == 4 ? a.sig_c[g] : a.rgb_c[g * 3 + (ch - 1)];
  }
  if (i < N) {
    const size_t g = (size_t)ray * a.Nf + (i - a.Nc);
    return ch == 0 ? a.t_f[g] : ch == 4 ? a.sig_f[g] : a.rgb_f[g * 3 + (ch - 1)];
  }
  return __builtin_nanf("");
}
// sigma / rgb of the workgroup's OWN fine samples as the bf16 training forward leaves them in LDS for its epilogue: wave w's 32 samples in a
// 2-KiB slot of their own (sigma [32] at +0, rgb [32][3] at +128 bytes); s = sample index inside the workgroup
__device__ __forceinline__ float park_sigma(const unsigned char* park, int s) { return reinterpret_cast<const float*>(park + 2048 * (s >> 5))[s & 31]; }
__device__ __forceinline__ float park_rgb(const unsigned char* park, int s, int c) { return reinterpret_cast<const float*>(park + 2048 * (s >> 5) + 128)[3 * (s & 31) + c]; }

// ONE (ray, channel) sort job of k_merge<true> at P = 256 by one wave: load, sort with slots, leave the sorted channel in val [256] / idx [256]
// park != null: the fine samples' sigma / rgb come from the workgroup's LDS slots (s0 = the ray's first sample inside the workgroup) instead of
// a.sig_f / a.rgb_f -- the same values, without waiting for their stores
__device__ __forceinline__ void merge_channel_job(const MergeArgs& a, const int ray, const int ch, const int lane, float* val, uint16_t* idx,
                                                  const unsigned char* park = nullptr, const int s0 = 0) {
  float v[4];
  unsigned ix[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * lane + r;
    if (park && ch != 0 && i >= a.Nc && i < a.Nc + a.Nf) v[r] = ch == 4 ? park_sigma(park, s0 + i - a.Nc) : park_rgb(park, s0 + i - a.Nc, ch - 1);
    else v[r] = merge_slot_value(a, ray, ch, i);
  }
  sort256_one_channel_idx(v, ix, lane);
  *reinterpret_cast<float4*>(val + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
  uint2 pk;
  pk.x = ix[0] | (ix[1] << 16);
  pk.y = ix[2] | (ix[3] << 16);
  *reinterpret_cast<uint2*>(idx + 4 * lane) = pk;
}

template <bool WITH_IDX>
__device__ __forceinline__ void merge_ray_composite(const float* val, const uint16_t* idx, int P, int N, float last, int lane, float* w_out,
                                                    float* bundle_out, uint16_t* perm_out, float* C_out, float* c_ret = nullptr);

// nerf.py:302-321 behind the load: val [5][P] (channel 0 = t, 1..3 = rgb, 4 = sigma; slots >= N padded with NaN) and, WITH_IDX, idx [5][P]
// = the original slot, in LDS -> five independent ascending channel sorts (quirk Q1), delta_i = t_{i+1} - t_i with the last = `last`,
// weights, C_fine[3]; optionally w [N], the sorted bundle [N][5] and the permutations perm [5][N] (global).
struct MergeNoFix { __device__ __forceinline__ void operator()() const {} };
template <bool WITH_IDX, class Sync, class PostSort = MergeNoFix>
__device__ __forceinline__ void merge_ray_sort_composite(float* val, uint16_t* idx, int P, int N, float last, int lane, float* w_out, float* bundle_out,
                                                         uint16_t* perm_out, float* C_out, Sync&& sync, PostSort&& post_sort = MergeNoFix{}) {
  if (P == 256) {  // the usual size (64 + 128 samples): the whole network in registers
    sort256_regs<WITH_IDX>(val, idx, lane, sync);
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
      sync();
    }
  }
  post_sort();  // (the joint-sort mode re-fills channels 1..4 by the depth channel's permutation here)
  merge_ray_composite<WITH_IDX>(val, idx, P, N, last, lane, w_out, bundle_out, perm_out, C_out);
}

// the composite over the SORTED channels val [5][P] (and idx): delta_i = t_{i+1} - t_i with the last = `last`, weights, C_fine[3]
template <bool WITH_IDX>
__device__ __forceinline__ void merge_ray_composite(const float* val, const uint16_t* idx, int P, int N, float last, int lane, float* w_out,
                                                    float* bundle_out, uint16_t* perm_out, float* C_out, float* c_ret) {
  double carry = 0.0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    const bool v = i < N;
    const float ti = v ? val[i] : 0.f;
    const float dl = (v && i + 1 < N) ? (val[i + 1] - ti) : last;
    const float sg = v ? val[4 * P + i] : 0.f;
    const float s = v ? dl * sg : 0.f;
    double cs = wave_incl_scan((double)s, lane) + carry;
    carry = __shfl(cs, 63);
    const float T = expf(-(float)cs);
    const float wi = v ? T * (1.0f - expf(-s)) : 0.f;
    if (v) {
      const float r = val[P + i], g = val[2 * P + i], b = val[3 * P + i];
      c0 += wi * r; c1 += wi * g; c2 += wi * b;
      if (w_out) w_out[i] = wi;
      if (bundle_out) {
        float* o = bundle_out + (size_t)i * 5;
        o[0] = ti; o[1] = r; o[2] = g; o[3] = b; o[4] = sg;
      }
      if (WITH_IDX && perm_out) {
#pragma unroll
        for (int c = 0; c < 5; ++c) perm_out[(size_t)c * N + i] = idx[c * P + i];
      }
    }
  }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  if (lane == 0) {
    C_out[0] = c0;
    C_out[1] = c1;
    C_out[2] = c2;
  }
  if (c_ret) {  // (the sums are wave-uniform: every lane holds them)
    c_ret[0] = c0;
    c_ret[1] = c1;
    c_ret[2] = c2;
  }
}

// ---- the per-ray stages as their kernels run them (arguments, liveness, stores), for ONE ray by one wave: k_coarse / k_merge (ray_ops.hip)
// and the small-batch epilogues of the bf16 training forward kernels (field_fwd_bf16.hip) are these functions ----

// k_coarse for ray `ray_raw` (rays behind the batch: computed on a copy of the last ray, nothing stored); w / cdf / tc: Nc floats of LDS each
template <class Sync>
__device__ __forceinline__ void coarse_ray_stage(const CoarseArgs& a, const int ray_raw, const int lane, float* w, float* cdf, float* tc, Sync&& sync) {
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
  // spacing of RAY 0 used for every ray (quirk Q6, nerf.py:233): t[0][1] - t[0][0]
  float delta0 = a.delta0;
  if (a.delta0_mode == 0) {
    const float n0 = a.ray0_override ? a.near0 : a.rayf[RF_NEAR];
    const float f0 = a.ray0_override ? a.far0 : a.rayf[RF_FAR];
    delta0 = ray0_spacing(n0, f0, a.Nc);
  }
  const size_t g0 = (size_t)ray * a.Nc;
  float lo, hi;
  coarse_ray_weights(a.sigma + g0, a.rgb + g0 * 3, a.t_c + g0, near, far, a.Nc, lane, w, cdf, tc, (live && a.w_c) ? a.w_c + g0 : nullptr,
                     (live && a.C_coarse) ? a.C_coarse + (size_t)ray * 3 : nullptr, lo, hi);
  sync();
  const bool bad = coarse_ray_resample(w, cdf, tc, lo, hi, delta0, a.Nc, a.Nf, lane, live ? a.t_f + (size_t)ray * a.Nf : nullptr);
  if (live && bad && a.status) atomicOr(a.status, 1u);
  if (live && bad && a.sticky) atomicOr(a.sticky, 1u);
}

// k_merge for ray `ray` (< B): load the ray's coarse + fine samples into val [5][P] (idx [5][P]: original slots), sort, composite
template <bool WITH_IDX, class Sync>
__device__ __forceinline__ void merge_ray_stage(const MergeArgs& a, const int ray, const int lane, float* val, uint16_t* idx, Sync&& sync) {
  const int P = a.P, N = a.Nc + a.Nf;
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
  sync();
  const size_t gN = (size_t)ray * N;
  // NERF_HIP_CORRECTED (a.joint; WITH_IDX only): the five channels were sorted independently above like the reference's (quirk Q1); the joint
  // mode keeps the DEPTH channel's stable sort and lets its permutation carry rgb and sigma along -- the other four sorts are overwritten
  auto joint_fix = [&]() {
    if constexpr (WITH_IDX) {
      if (a.joint) {
        for (int i = lane; i < P; i += 64) {
          const int slot = idx[i];
#pragma unroll
          for (int c = 1; c < 5; ++c) {
            val[c * P + i] = merge_slot_value(a, ray, c, slot);
            idx[c * P + i] = (uint16_t)slot;
          }
        }
        sync();
      }
    }
  };
  merge_ray_sort_composite<WITH_IDX>(val, idx, P, N, a.last, lane, a.w ? a.w + gN : nullptr, a.bundle ? a.bundle + gN * 5 : nullptr,
                                     (WITH_IDX && a.perm) ? a.perm + (size_t)ray * 5 * N : nullptr, a.C_fine + (size_t)ray * 3, sync, joint_fix);
}

// ---- ray_loss (nerf.py:325-331) in two pieces, bit-compatible with k_ray_loss (ray_ops.hip) ----
// element (ray, ch): e1 = C_c - C*, e2 = C_f - C*; d loss / d C_c = 2 e1, d loss / d C_f = 2 e2; summand = e1 e1 + e2 e2
__device__ __forceinline__ void ray_loss_element(float cc, float cf, float ct, float& dcc, float& dcf, float& term) {
  const float e1 = cc - ct, e2 = cf - ct;
  dcc = 2.0f * e1;
  dcf = 2.0f * e2;
  term = e1 * e1 + e2 * e2;
}
// the sum of n summands in k_ray_loss's order -- 1,024 virtual threads each adding its strided elements, a butterfly over each of the 16
// virtual waves, the 16 wave sums added in order -- by a workgroup of T = 256 or 512 threads (every thread plays 1024 / T virtual ones);
// red: 16 floats of LDS; every thread of the workgroup calls it, thread 0 stores the result
template <int T>
__device__ __forceinline__ void ray_loss_sum(const float* terms, int n, float* loss, float* red, int tid) {
  static_assert(1024 % T == 0 && T % 64 == 0, "whole virtual waves per real wave");
#pragma unroll
  for (int v = 0; v < 1024 / T; ++v) {
    const int vt = tid + v * T;  // virtual thread id
    float acc = 0.f;
    for (int i = vt; i < n; i += 1024) acc += terms[i];
    acc = wave_sum(acc);
    if ((vt & 63) == 0) red[vt >> 6] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    loss[0] = s;
  }
}

}  // namespace nerf
