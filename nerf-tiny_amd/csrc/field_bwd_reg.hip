// field_bwd_reg.hip -- the backward dX chain with REGISTER-RESIDENT gradients (MI355X / gfx950).
//
// Same arithmetic and the same packed (transposed) weight image as k_field_bwd (field_bwd.hip), organised like
// k_field_fwd_reg: one wave owns 32 samples and all 256 features; the raw accumulators of one layer, masked by the
// forward pass's ReLU bits, ARE the B operand of the next (transposed) layer, so there is no LDS and no barrier.
// Every layer's pre-activation gradient is streamed to the row-major G buffers by its CONSUMER, one 16-byte group per
// lane and k-block tile, interleaved with the MFMA stream (the weight-gradient GEMMs read G afterwards).
// FINE additionally carries d loss/d gamma_p (skip layer + layer 0) -> d loss/d point -> d loss/d t_fine (quirk Q9).
#include "field_common.h"

namespace nerf {

constexpr int RMB = 32;  // samples per wave

template <int NFT>
struct WStageB {
  float4 w[NFT];
};

__device__ __forceinline__ float f4cb(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// mask word of input tile t of one layer for this lane (forward kernel's layout: [layer][tile64][st][h*32 + j], entry (f, wv))
__device__ __forceinline__ unsigned mask_word(const uint16_t* __restrict__ mlayer, int t) { return mlayer[((t & 1) * 2) * 256 + (t >> 1) * 64]; }

// acc[f] (+)= sum_k Wt[f-tile][k] * in[k], in = prev tile values (MASK_IN: zeroed where the forward activation was not > 0).
// Mask words are fetched LAZILY, one input tile (4 k-blocks = 8192 cycles) ahead of their use: `mfirst` holds the word of
// tile 0 on entry; while tile t is activated the word of tile t+1 is requested, and during the last tile the word of tile 0
// of the NEXT masked layer (`mnext`; HAS_NEXT_MASK = false: there is none) is requested into `mfirst` -- two live registers instead of eight, and no
// load whose latency the wave has to sit out.  grow != nullptr: the activated input (= this layer's pre-activation gradient)
// is stored to the G rows (lanes past the end of the pass own a dump row, so the stores carry no predicate).
// Two fragment stages as in k_field_fwd_reg; st0 = k-block 0 on entry / next segment's k-block 0 on exit.
template <int KB, int NFT, int NKB, int NNFT, bool ZERO_INIT, bool MASK_IN, bool STORE = true, bool HAS_NEXT_MASK = MASK_IN>
__device__ __forceinline__ void reg_layer_bwd(const int seg, const int next_seg /* float4 offsets into the packed image; < 0: none */, int lane,
                                              const f32x16* prev, f32x16* acc, WStageB<8>& st0, const uint16_t* __restrict__ mlayer,
                                              const uint16_t* __restrict__ mnext, unsigned& mfirst, float* __restrict__ grow, const RegBuf& rb) {
  constexpr int KT = KB / 4;
  WStageB<8> st1;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 tin[2];
  unsigned mw = mfirst;  // word of the tile being activated
  auto activate = [&](int t) {
    unsigned mn = 0;
    if (MASK_IN) {
      if (t + 1 < KT) mn = mask_word(mlayer, t + 1);
      else if (mnext != nullptr) mfirst = mask_word(mnext, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tin[t & 1][r] = MASK_IN ? (((mw >> r) & 1u) ? prev[t][r] : 0.f) : prev[t][r];
    // Runtime pointer tests on purpose: measured, the uniform branches they leave in the stream make the kernel FASTER than
    // compile-time flags do (fine pass 5.10 vs 5.46 ms at cfg2) -- they cut the 9,000-MFMA stream into scheduling regions the
    // compiler handles better than one huge block.  A tile that is not stored is pinned to this program point instead.
    if (grow == nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(tin[t & 1][r]));
    }
    if (grow != nullptr) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store_row4(grow + 32 * t + 8 * g, make_float4(tin[t & 1][4 * g], tin[t & 1][4 * g + 1], tin[t & 1][4 * g + 2], tin[t & 1][4 * g + 3]));
    }
    if (MASK_IN) mw = mn;
  };
  activate(0);
  // one k-block: request the fragments of k-block kb+1 (or of the next segment) into `ld`, multiply with `cur`
  auto kblock = [&](int kb, const WStageB<8>& cur, WStageB<8>& ld) {
    if (kb + 1 < KB) {
#pragma unroll
      for (int f = 0; f < NFT; ++f) ld.w[f] = reg_ldw(rb, seg + (int)(f * KB + kb + 1) * 64);
    } else if (next_seg >= 0) {
#pragma unroll
      for (int f = 0; f < NNFT; ++f) ld.w[f] = reg_ldw(rb, next_seg + (int)(f * NKB) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((kb & 3) == 2 && (kb >> 2) + 1 < KT) activate((kb >> 2) + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float b = tin[(kb >> 2) & 1][4 * (kb & 3) + s];
#pragma unroll
      for (int f = 0; f < NFT; ++f) {
        if (ZERO_INIT && kb == 0 && s == 0)
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(cur.w[f], s), b, zero, 0, 0, 0);
        else
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(cur.w[f], s), b, acc[f], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int kb = 0; kb < KB; kb += 2) {  // KB is even for every segment
    kblock(kb, st0, st1);
    kblock(kb + 1, st1, st0);
  }
}

// The 2-tile segments (d gamma_p: layer 4's skip columns and layer 0, fine pass only): 8 MFMAs per k-block instead of 32, so a
// fragment requested one k-block ahead arrives 512 cycles later -- less than an L2 round trip, and the wave sat out a part of
// it on every k-block (stamped: 47 k cycles for 8 k of MFMA in layer 0).  Here the 16 fragment registers of the two stages
// form a ring of 8 k-blocks: requests run 8 k-blocks (4096 cycles) ahead.  st0.w[0..1] = k-block 0 on entry; on exit st0
// holds the next segment's k-block 0 (NNFT tiles), requested at k-block KB - 4 when st0's half of the ring has drained.
template <int KB, int NKB, int NNFT, bool ZERO_INIT, bool STORE, bool HAS_NEXT_MASK>
__device__ __forceinline__ void reg_layer_bwd_thin(const int seg, const int next_seg /* float4 offsets into the packed image; < 0: none */, int lane,
                                                   const f32x16* prev, f32x16* acc, WStageB<8>& st0, const uint16_t* __restrict__ mlayer,
                                                   const uint16_t* __restrict__ mnext, unsigned& mfirst, float* __restrict__ grow, const RegBuf& rb) {
  static_assert(KB == 32, "ring indexing below assumes 32 k-blocks");
  constexpr int KT = KB / 4, D = 8;
  WStageB<8> st1;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 tin[2];
  unsigned mw = mfirst;
  auto slot = [&](int kb, int f) -> float4& { return ((kb % D) < 4) ? st0.w[2 * (kb % 4) + f] : st1.w[2 * (kb % 4) + f]; };
  auto activate = [&](int t) {
    unsigned mn = 0;
    if (t + 1 < KT) mn = mask_word(mlayer, t + 1);
    else if (HAS_NEXT_MASK) mfirst = mask_word(mnext, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) tin[t & 1][r] = ((mw >> r) & 1u) ? prev[t][r] : 0.f;
    if (STORE) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store_row4(grow + 32 * t + 8 * g, make_float4(tin[t & 1][4 * g], tin[t & 1][4 * g + 1], tin[t & 1][4 * g + 2], tin[t & 1][4 * g + 3]));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(tin[t & 1][r]));
    }
    mw = mn;
  };
  // fill the ring: k-blocks 1 .. D-1 (k-block 0 came with the previous segment)
#pragma unroll
  for (int kb = 1; kb < D; ++kb)
#pragma unroll
    for (int f = 0; f < 2; ++f) slot(kb, f) = reg_ldw(rb, seg + (int)(f * KB + kb) * 64);
  activate(0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if ((kb & 3) == 2 && (kb >> 2) + 1 < KT) activate((kb >> 2) + 1);
    const float4 w0 = slot(kb, 0), w1 = slot(kb, 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float b = tin[(kb >> 2) & 1][4 * (kb & 3) + s];
      if (ZERO_INIT && kb == 0 && s == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(w0, s), b, zero, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(w1, s), b, zero, 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(w0, s), b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4cb(w1, s), b, acc[1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the slot just consumed takes k-block kb + D; once the ring's st0 half has drained for good it takes the next segment
    if (kb + D < KB) {
#pragma unroll
      for (int f = 0; f < 2; ++f) slot(kb, f) = reg_ldw(rb, seg + (int)(f * KB + kb + D) * 64);
    } else if (kb == KB - 4 && next_seg >= 0) {
#pragma unroll
      for (int f = 0; f < NNFT; ++f) st0.w[f] = reg_ldw(rb, next_seg + (int)(f * NKB) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef NERF_STAMPS  // diagnostic build only (make stamps): cycle sums per phase, see scripts/phase_stamps.py
#define BSTAMP(slot)                                            \
  do {                                                          \
    __builtin_amdgcn_sched_barrier(0);                          \
    const unsigned long long t_ = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                          \
    tsum[slot] += t_ - tlast;                                   \
    tlast = t_;                                                 \
  } while (0)
#else
#define BSTAMP(slot) do { } while (0)
#endif

template <bool FINE>
__global__ __launch_bounds__(64, 1) void k_field_bwd_reg(const FieldBwdArgs a) {
#ifdef NERF_STAMPS
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * RMB;
  const int m = m0 + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const size_t MS = (size_t)a.MSrows * WIDTH;
  // this lane's row in the saved tensors / gradient buffers: lanes past the end of the pass own dump row Mtot + j (their
  // gradients are exact zeros -- dz = ds = 0 below -- so what they store is harmless and what they read only has to be finite)
#ifdef NERF_TIMING_SAVE_ALIAS  // (timing experiments only: saved inputs read from / gradients written to the dump rows)
  const long long rrow = a.Mtot + j;
#else
  const long long rrow = valid ? (long long)(a.row0 + m) : a.Mtot + j;
#endif
  const size_t grow_off = (size_t)rrow * WIDTH + 4 * h;  // this lane's 16-byte groups start here
  float* const grow = a.G + grow_off;
  const float* const srow = a.save + grow_off;
  const uint16_t* const mrow = a.masks + ((size_t)(a.tile0 + (m0 >> 6)) * 4 + ((m0 >> 5) & 1)) * 256 + h * 32 + j;
  const size_t MKS = (size_t)a.tiles_tot * 4 * 256;
  const RegBuf rb = reg_buf(a.wp, threadIdx.x);  // (64-thread workgroups: threadIdx.x is the lane)
  constexpr int L256 = 8 * 32 * 64;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  WStageB<8> st0;
#pragma unroll
  for (int f = 0; f < 8; ++f) st0.w[f] = reg_ldw(rb, seg_off4(SEG_T_FOLD) + (f * 16) * 64);
  auto mlayer = [&](int layer) { return mrow + (size_t)layer * MKS; };
  unsigned mfirst = mask_word(mlayer(7), 0);  // first masked layer of the chain; every later word is fetched a tile ahead

  // ---- colour head backward (VALU): rgb = sigmoid(z), z = W_c c + b, c = relu(pre_d)  ->  dpre_d in registers.
  // All loads first (the 16 c groups come from HBM): left to itself the compiler sinks the weight loads into per-element
  // branches and the wave sits out sixteen dependent round trips before the first MFMA.
  f32x16 D0[4];
  float ds;  // d loss / d sigma_pre of this lane's sample
  {
    const float* crow = srow + S_C * MS;
    float4 cv[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) cv[t][g] = *reinterpret_cast<const float4*>(crow + 32 * t + 8 * g);
    float dz[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float o = a.rgb[(size_t)mc * 3 + ch];
      const float up = a.drgb[(size_t)mc * 3 + ch];
      dz[ch] = valid ? up * ((1.0f - o) * o) : 0.f;
    }
    // sigma head upstream: sigma = |pre|, d|x|/dx with sign(0) = 0 like torch.  (dz_r, dz_g, dz_b, dsigma_pre) is the A operand
    // of the thin-heads weight-gradient product (dw_f32.hip)
    const float sp = a.spre[a.row0 + mc];
    const float sgn = sp > 0.f ? 1.0f : (sp < 0.f ? -1.0f : 0.f);
    ds = valid ? a.dsig[mc] * sgn : 0.f;
    if (valid && h == 0) {
      *reinterpret_cast<float4*>(a.dz + (size_t)(a.row0 + m) * 4) = make_float4(dz[0], dz[1], dz[2], ds);
      a.dspre[a.row0 + m] = ds;
    }
    const float* wc = a.w.p[W_COLOR] + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 q0[4], q1[4], q2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        q0[g] = *reinterpret_cast<const float4*>(wc + 32 * t + 8 * g);
        q1[g] = *reinterpret_cast<const float4*>(wc + HALF + 32 * t + 8 * g);
        q2[g] = *reinterpret_cast<const float4*>(wc + 2 * HALF + 32 * t + 8 * g);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = __builtin_fmaf(q2[g].x, dz[2], __builtin_fmaf(q1[g].x, dz[1], q0[g].x * dz[0]));
        const float v1 = __builtin_fmaf(q2[g].y, dz[2], __builtin_fmaf(q1[g].y, dz[1], q0[g].y * dz[0]));
        const float v2 = __builtin_fmaf(q2[g].z, dz[2], __builtin_fmaf(q1[g].z, dz[1], q0[g].z * dz[0]));
        const float v3 = __builtin_fmaf(q2[g].w, dz[2], __builtin_fmaf(q1[g].w, dz[1], q0[g].w * dz[0]));
        D0[t][4 * g + 0] = cv[t][g].x > 0.f ? v0 : 0.f;
        D0[t][4 * g + 1] = cv[t][g].y > 0.f ? v1 : 0.f;
        D0[t][4 * g + 2] = cv[t][g].z > 0.f ? v2 : 0.f;
        D0[t][4 * g + 3] = cv[t][g].w > 0.f ? v3 : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  BSTAMP(0);  // prologue: colour head backward
  f32x16 A[8], B[8];
  // ---- dir_info and point_info backward as ONE transposed layer (common.h SEG_FOLD), + the sigma head:
  // dh7 = W_fold^T dpre_d + w_sigma (x) dsigma_pre  (256 <- 128); stores dpre_d
  {
    const float dsb = (h == 0) ? ds : 0.f;  // outer product as one MFMA per tile: A = w_sigma rows, B = ds on lane half 0
    const float* ws = a.w.p[W_SIGMA];
#pragma unroll
    for (int f = 0; f < 8; ++f) B[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[f * 32 + j], dsb, zero, 0, 0, 0);
  }
  reg_layer_bwd<16, 8, 32, 8, false, false>(seg_off4(SEG_T_FOLD), seg_off4(SEG_T_L7), lane, D0, B, st0, nullptr, nullptr, mfirst,
                                            grow + G_D * MS, rb);
  BSTAMP(1);  // dir_info + point_info folded, sigma head (520 MFMAs)
  BSTAMP(2);
  // ---- layers 7, 6, 5: input = raw d h_l masked by h_l > 0 (= dpre_l, stored), output = raw d h_{l-1}
  constexpr int sT7 = seg_off4(SEG_T_L7);
  reg_layer_bwd<32, 8, 32, 8, true, true>(sT7, sT7 + L256, lane, B, A, st0, mlayer(7), mlayer(6), mfirst, grow + 7 * MS, rb);
  reg_layer_bwd<32, 8, 32, 8, true, true>(sT7 + L256, sT7 + 2 * L256, lane, A, B, st0, mlayer(6), mlayer(5), mfirst, grow + 6 * MS, rb);
  reg_layer_bwd<32, 8, 32, 8, true, true>(sT7 + 2 * L256, seg_off4(SEG_T_L4A), lane, B, A, st0, mlayer(5), mlayer(4), mfirst, grow + 5 * MS, rb);
  BSTAMP(3);  // layers 7..5 (3,072 MFMAs)
  // ---- layer 4 (input cat(h3, gamma_p)): d h3, and for the fine pass d gamma_p through the skip connection
  f32x16 accg[2];
  if (FINE) {
    reg_layer_bwd<32, 8, 32, 2, true, true>(seg_off4(SEG_T_L4A), seg_off4(SEG_T_L4B), lane, A, B, st0, mlayer(4), mlayer(4), mfirst, grow + 4 * MS, rb);
    reg_layer_bwd_thin<32, 32, 8, true, false, true>(seg_off4(SEG_T_L4B), seg_off4(SEG_T_L3), lane, A, accg, st0, mlayer(4), mlayer(3), mfirst, nullptr, rb);
  } else {
    reg_layer_bwd<32, 8, 32, 8, true, true>(seg_off4(SEG_T_L4A), seg_off4(SEG_T_L3), lane, A, B, st0, mlayer(4), mlayer(3), mfirst, grow + 4 * MS, rb);
  }
  BSTAMP(4);  // layer 4 (1,024 / 1,280 MFMAs)
  // ---- layers 3, 2, 1
  constexpr int sT3 = seg_off4(SEG_T_L3);
  reg_layer_bwd<32, 8, 32, 8, true, true>(sT3, sT3 + L256, lane, B, A, st0, mlayer(3), mlayer(2), mfirst, grow + 3 * MS, rb);
  reg_layer_bwd<32, 8, 32, 8, true, true>(sT3 + L256, sT3 + 2 * L256, lane, A, B, st0, mlayer(2), mlayer(1), mfirst, grow + 2 * MS, rb);
  unsigned mb0[8];  // coarse pass: the eight mask words of layer 0 for the epilogue, requested a whole layer ahead
  if (!FINE) {
#pragma unroll
    for (int t = 0; t < 8; ++t) mb0[t] = mask_word(mlayer(0), t);
  }
  reg_layer_bwd<32, 8, 32, 2, true, true, true, FINE>(sT3 + 2 * L256, FINE ? seg_off4(SEG_T_L0) : -1, lane, B, A, st0, mlayer(1),
                                                      mlayer(0), mfirst, grow + 1 * MS, rb);
  BSTAMP(5);  // layers 3..1 (3,072 MFMAs)
  // ---- dpre_0 = d h0 masked; fine: d gamma_p += W_0^T dpre_0
  if (FINE) {
    reg_layer_bwd_thin<32, 32, 2, false, true, false>(seg_off4(SEG_T_L0), -1, lane, A, accg, st0, mlayer(0), nullptr, mfirst, grow, rb);
    // gamma -> point -> depth.  accg[t][4g + 2e], [.. + 1] = d loss / d (sin, cos) of pair pi = 4(4t+g) + 2h + e.
    // d gamma / d x needs (cos, -sin) of the same phases: they ARE the saved layer-0 input (tensor S_GP: this lane's eight
    // 16-byte groups hold exactly its pairs), so they are loaded, not recomputed -- 8 loads instead of ~15 sincos with their
    // float64 argument reductions behind the last MFMA, where nothing overlaps them.
    // (this lane's row and ray are re-derived from the lane id behind the stream -- mbcnt, not threadIdx: nothing of the prologue
    // stays alive across 8,200 MFMAs, where the allocator used to park four address registers in scratch)
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int j_e = lane_e & 31, h_e = lane_e >> 5;
    const int m_e = blockIdx.x * RMB + j_e;
    const bool valid_e = m_e < a.M;
    const int mc_e = valid_e ? m_e : a.M - 1;
    const long long rrow_e = valid_e ? (long long)(a.row0 + m_e) : a.Mtot + j_e;
    const float* rf = a.rayf + (size_t)(mc_e / a.N) * RAYF;
    const float* gprow = a.save + (size_t)rrow_e * WIDTH + 4 * h_e + S_GP * ((size_t)a.MSrows * WIDTH);
    float4 gq[8];
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) gq[g8] = *reinterpret_cast<const float4*>(gprow + 8 * g8);
    float dp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pi = 4 * g8 + 2 * h_e + e;
        if (pi < 30) {
          const int c = pi / 10, l = pi - 10 * c;
          const float fl = __uint_as_float(kFreqPointBits[l]);
          const float sn = e == 0 ? gq[g8].x : gq[g8].z, cn = e == 0 ? gq[g8].y : gq[g8].w;
          const float dgs = accg[g8 >> 2][4 * (g8 & 3) + 2 * e], dgc = accg[g8 >> 2][4 * (g8 & 3) + 2 * e + 1];
          const float contrib = fl * (cn * dgs - sn * dgc);
          if (c == 0) dp[0] += contrib; else if (c == 1) dp[1] += contrib; else dp[2] += contrib;
        }
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) dp[c] += __shfl_xor(dp[c], 32);
    if (valid_e && h_e == 0) {
      const float dtp = __builtin_fmaf(rf[RF_DWRD + 2], dp[2], __builtin_fmaf(rf[RF_DWRD + 1], dp[1], rf[RF_DWRD] * dp[0]));
      a.dt[m_e] += dtp;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = ((mb0[t] >> (4 * g + 0)) & 1u) ? A[t][4 * g + 0] : 0.f;
        v.y = ((mb0[t] >> (4 * g + 1)) & 1u) ? A[t][4 * g + 1] : 0.f;
        v.z = ((mb0[t] >> (4 * g + 2)) & 1u) ? A[t][4 * g + 2] : 0.f;
        v.w = ((mb0[t] >> (4 * g + 3)) & 1u) ? A[t][4 * g + 3] : 0.f;
        store_row4(grow + 32 * t + 8 * g, v);
      }
  }
#ifdef NERF_STAMPS
  BSTAMP(6);  // layer 0 / epilogue
  if (a.stamps && lane == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(a.stamps + (FINE ? 8 : 0) + i, tsum[i]);
    atomicAdd(a.stamps + (FINE ? 30 : 31), 1ull);
  }
#endif
}

hipError_t launch_field_bwd_reg(const FieldBwdArgs& a, bool fine, hipStream_t st) {
  const int tiles = (a.M + RMB - 1) / RMB;
  if (fine)
    hipLaunchKernelGGL(k_field_bwd_reg<true>, dim3(tiles), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL(k_field_bwd_reg<false>, dim3(tiles), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace nerf
