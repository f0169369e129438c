// field_common.h -- device building blocks shared by the fused field kernels (forward and backward chain).
#pragma once
#include "kernels.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// 16-byte store of a saved activation / gradient group of the fp32 training kernels (written once, read once by the weight-gradient
// kernels, 8 GB per step) with the non-temporal hint: measured 1.1 % on the fp32 train step (19.37 -> 19.16 ms; the fine chain
// 4.50 -> 4.36 ms) against plain stores -- as for the bf16 saves (bf16_stream.h), less dirty data parked in L2.
__device__ __forceinline__ void store_row4(float* dst, const float4& v) {
#ifdef NERF_F32_PLAIN_SAVES  // (timing experiment)
  *reinterpret_cast<float4*>(dst) = v;
#else
  const f32x4v q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<f32x4v*>(dst));
#endif
}

// Weight-fragment loads of the register-streaming kernels (field_fwd_reg.hip, field_bwd_reg.hip) as BUFFER loads: address = resource
// base (the packed image: 4 SGPRs) + this lane's 16 bytes (ONE VGPR for the whole kernel) + the fragment's byte offset as the
// instruction's scalar offset.  No vector address arithmetic at all -- with `global_load` the compiler formed a new 64-bit vector address
// (v_add_co + v_addc, their hazard nops, a dependent load) for every ~4 fragments, 1,080 VALU instructions per 32-sample tile, and on
// this chip every VALU instruction between fp32 MFMAs is matrix time: the inference kernel went from 2.98 to 2.83 ms per launch (-5 %).
// The compiler still sees loads, so its counted `s_waitcnt vmcnt` stay.
typedef unsigned rb_u32x4 __attribute__((ext_vector_type(4)));
struct RegBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  int lane16;
};
__device__ __forceinline__ RegBuf reg_buf(const float4* wp, int lane) {
  return RegBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wp), 0, PACKED_ALL_F4 * 16, 0x00020000), lane * 16};
}
// 16 bytes per lane at float4 index `f4` of the packed image (wave-uniform: segment offset + fragment * 64; a literal wherever the
// stream is unrolled).  An int on purpose: with pointers the compiler lost the uniformity of the difference in one kernel and wrapped
// every load in a waterfall loop.
__device__ __forceinline__ float4 reg_ldw(const RegBuf& rb, int f4) {
  int soff = __builtin_amdgcn_readfirstlane(f4 * 16);  // (a literal almost everywhere; where the compiler keeps a loop counter in a VGPR: one v_readfirstlane)
  asm("" : "+s"(soff));  // stays a scalar operand: left alone the compiler folds the literal into the VECTOR offset (an add per
  //                                 fragment again) or, worse, moves the lane part to the scalar side behind a waterfall loop
  const rb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb.rsrc, rb.lane16, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ------------------------------------------------------------------------------------------
// building blocks of the fused kernels
// ------------------------------------------------------------------------------------------

// A fragments of k-blocks 0 and 1 of a packed segment for this wave's NFT feature tiles; issued EARLY (before the
// barriers / epilogue of the previous layer) so that the L2 latency of a layer's first loads hides behind them
template <int NFT> struct WFrag { float4 c[NFT]; float4 n[NFT]; };

template <int KB, int NFT>
__device__ __forceinline__ void wfrag_first(const float4* __restrict__ wseg, int ft0, int lane, WFrag<NFT>& w) {
  const float4* wbase = wseg + (size_t)ft0 * KB * 64 + lane;
#pragma unroll
  for (int f = 0; f < NFT; ++f) w.c[f] = wbase[(size_t)f * KB * 64];
#pragma unroll
  for (int f = 0; f < NFT; ++f) w.n[f] = wbase[(size_t)(f * KB + 1) * 64];
}

// acc[f][st] += W[ft0+f tile][k] * act[st*32 + sample][k]  for k in [0, 8*KB); w = fragments of k-blocks 0, 1 (wfrag_first).
// The A fragments of k-block kb+2 are requested before the MFMAs of k-block kb (two k-blocks = 2048+ cycles of MFMA
// cover the L2 latency also when the partner wave on the SIMD is stalled).
template <int KB, int NFT>
__device__ __forceinline__ void mfma_layer(const float4* __restrict__ wseg, int ft0, const float* act, int kcol0, int lane,
                                           f32x16 (&acc)[NFT][2], WFrag<NFT>& w) {
  const int j = lane & 31, h = lane >> 5;
  const float* a0p = act + j * LDA + kcol0 + 4 * h;
  const float* a1p = a0p + 32 * LDA;
  const float4* wbase = wseg + (size_t)ft0 * KB * 64 + lane;
  float4 w2[NFT];
#pragma unroll 2
  for (int kb = 0; kb < KB; ++kb) {
    const int k2 = (kb + 2 < KB) ? kb + 2 : KB - 1;
#pragma unroll
    for (int f = 0; f < NFT; ++f) w2[f] = wbase[(size_t)(f * KB + k2) * 64];
    const float4 a0 = *reinterpret_cast<const float4*>(a0p + kb * 8);
    const float4 a1 = *reinterpret_cast<const float4*>(a1p + kb * 8);
#pragma unroll
    for (int f = 0; f < NFT; ++f) {
      acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].x, a0.x, acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].x, a1.x, acc[f][1], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < NFT; ++f) {
      acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].y, a0.y, acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].y, a1.y, acc[f][1], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < NFT; ++f) {
      acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].z, a0.z, acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].z, a1.z, acc[f][1], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < NFT; ++f) {
      acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].w, a0.w, acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.c[f].w, a1.w, acc[f][1], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < NFT; ++f) {
      w.c[f] = w.n[f];
      w.n[f] = w2[f];
    }
  }
}

// convenience form: loads its own first fragments
template <int KB, int NFT>
__device__ __forceinline__ void mfma_layer(const float4* __restrict__ wseg, int ft0, const float* act, int kcol0, int lane,
                                           f32x16 (&acc)[NFT][2]) {
  WFrag<NFT> w;
  wfrag_first<KB, NFT>(wseg, ft0, lane, w);
  mfma_layer<KB, NFT>(wseg, ft0, act, kcol0, lane, acc, w);
}

// bias values of this lane's accumulator rows, requested EARLY (like wfrag_first) and applied by acc_init_regs
template <int NFT>
__device__ __forceinline__ void bias_first(const float* __restrict__ bias, int fbase, int lane, float4 (&bq)[NFT][4]) {
  const int h = lane >> 5;
#pragma unroll
  for (int f = 0; f < NFT; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[f][g] = *reinterpret_cast<const float4*>(bias + fbase + f * 32 + 8 * g + 4 * h);
}
template <int NFT>
__device__ __forceinline__ void acc_init_regs(const float4 (&bq)[NFT][4], f32x16 (&acc)[NFT][2]) {
#pragma unroll
  for (int f = 0; f < NFT; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        acc[f][st][4 * g + 0] = bq[f][g].x;
        acc[f][st][4 * g + 1] = bq[f][g].y;
        acc[f][st][4 * g + 2] = bq[f][g].z;
        acc[f][st][4 * g + 3] = bq[f][g].w;
      }
}

// accumulators <- bias (feature = fbase + f*32 + 8g + 4h + r for register 4g + r)
template <int NFT>
__device__ __forceinline__ void acc_init_bias(const float* __restrict__ bias, int fbase, int lane, f32x16 (&acc)[NFT][2]) {
  const int h = lane >> 5;
#pragma unroll
  for (int f = 0; f < NFT; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *reinterpret_cast<const float4*>(bias + fbase + f * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        acc[f][st][4 * g + 0] = b.x;
        acc[f][st][4 * g + 1] = b.y;
        acc[f][st][4 * g + 2] = b.z;
        acc[f][st][4 * g + 3] = b.w;
      }
    }
}

// accumulators -> LDS activation rows (optionally through ReLU).  mask (training): one uint16 per (f, st) tile and
// lane, bit 4g+r = accumulator register 4g+r was > 0; stored at mask[(f*2+st)*256] (caller offsets by thread).
template <int NFT, bool RELU>
__device__ __forceinline__ void acc_store(float* act, int fbase, int lane, const f32x16 (&acc)[NFT][2], uint16_t* mask = nullptr) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int f = 0; f < NFT; ++f)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (RELU && mask) {
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= (acc[f][st][r] > 0.f) ? (1u << r) : 0u;
        mask[(f * 2 + st) * 256] = (uint16_t)bits;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(acc[f][st][4 * g], acc[f][st][4 * g + 1], acc[f][st][4 * g + 2], acc[f][st][4 * g + 3]);
        if (RELU) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f);
          v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(act + (st * 32 + j) * LDA + fbase + f * 32 + 8 * g + 4 * h) = v;
      }
    }
}

// world point of a sample: v = d_cam * t; p = ((R0*v0 + R1*v1) + R2*v2) + o   (nerf.py:200-216; every product
// and sum rounded separately -- the translation unit is built with -ffp-contract=off)
__device__ __forceinline__ void sample_point(const float* __restrict__ rf, float t, float (&p)[3]) {
  const float v0 = rf[RF_DCAM + 0] * t, v1 = rf[RF_DCAM + 1] * t, v2 = rf[RF_DCAM + 2] * t;
#pragma unroll
  for (int c = 0; c < 3; ++c) p[c] = ((rf[RF_R + 3 * c] * v0 + rf[RF_R + 3 * c + 1] * v1) + rf[RF_R + 3 * c + 2] * v2) + rf[RF_O + c];
}

// sin and cos of a phase of magnitude up to ~1e6 rad, each within 1 ulp of the correctly rounded value on [-1, 1]:
// the argument is reduced in fp64 (r = ph - n*pi/2 with one fma: error < 1e-10 rad for |ph| < 2^20), then the cephes
// single-precision minimax polynomials on [-pi/4, pi/4] are evaluated in fp32 and the quadrant is applied.
// ~25 instructions instead of the ~150 (plus a Payne-Hanek loop) of the generic sinf + cosf pair.
__device__ __forceinline__ void sincos_phase(float ph, float& sn, float& cs) {
  const double x = (double)ph;
  const double n = __builtin_rint(x * 0.63661977236758134308);           // 2/pi
  const float r = (float)__builtin_fma(-n, 1.57079632679489661923, x);  // pi/2
  const int q = (int)((long long)n & 3);  // quadrant (long long: no overflow for any finite fp32 phase)
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * z, r, r);  // sin(r)
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));  // cos(r)
  const float a = (q & 1) ? pc : ps;
  const float b = (q & 1) ? ps : pc;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}

// gamma_p of this thread's sample -> act[sm][0..63] (cols 60..63 = 0).  Wave wv writes (c,l) pairs 8wv .. 8wv+7.
// gamma[c*20 + 2l + s] = (sin, cos)[s](fp32(x_c * f_l))   (nerf.py:135-167, flatten nerf.py:103)
__device__ __forceinline__ void encode_point_to_lds(const float (&p)[3], float* act, int sm, int wv) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = wv * 8 + i;  // wave-uniform
    float2 sc = make_float2(0.f, 0.f);
    int col;
    if (e < 30) {
      const int c = e / 10, l = e - 10 * c;
      const float x = (c == 0) ? p[0] : ((c == 1) ? p[1] : p[2]);
      const float ph = x * __uint_as_float(kFreqPointBits[l]);
      sincos_phase(ph, sc.x, sc.y);
      col = c * 20 + 2 * l;
    } else {
      col = 60 + 2 * (e - 30);
    }
    *reinterpret_cast<float2*>(act + sm * LDA + col) = sc;
  }
}

// copy act[0..63][0 .. 4*ncol4) to rows [grow0, grow0 + nrows) of a row-major global buffer with row stride ldd
__device__ __forceinline__ void save_rows(const float* act, float* __restrict__ dst, long long grow0, int nrows, int ncol4, int ldd, int tid) {
  for (int idx = tid; idx < TM * ncol4; idx += 256) {
    const int r = idx / ncol4, c4 = idx - r * ncol4;
    if (r < nrows) {
      const float4 v = *reinterpret_cast<const float4*>(act + r * LDA + 4 * c4);
      *reinterpret_cast<float4*>(dst + (size_t)(grow0 + r) * ldd + 4 * c4) = v;
    }
  }
}


}  // namespace nerf
