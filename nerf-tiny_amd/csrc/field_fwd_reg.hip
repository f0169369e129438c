// field_fwd_reg.hip -- inference forward of the field query with REGISTER-RESIDENT activations (MI355X / gfx950).
//
// Same arithmetic as k_field_fwd (field_fwd.hip) and the same packed weight image, but no LDS and no barriers:
//   * one 64-lane wave owns 32 samples and computes ALL 256 features of every layer for them:
//     D[feature][sample] = W . act  with v_mfma_f32_32x32x2_f32, 8 feature tiles x 16 accumulator VGPRs;
//   * in the 32x32 accumulator layout lane (j, h) holds features 32t + 8g + 4h + r of sample j.  The MFMA sums over
//     its two lane halves, and WHICH k each half supplies is free as long as A and B agree -- so the post-ReLU
//     accumulator registers of layer L are used, as they stand, as the B operand of layer L+1
//     (k-step (t, 4g + r): half h supplies k = 32t + 8g + 4h + r), and the A fragment that goes with it is the
//     4 consecutive weights W[i][32t + 8g + 4h .. +3] -- exactly the float4 the packed image already stores;
//   * nothing is shared between waves, so nothing synchronises: a wave streams 8,256 MFMAs per tile with a
//     2-stage register pipeline of A fragments (8 x 16-byte loads per k-block, L2/L1 resident, requested one
//     k-block = 32 MFMAs = 2048 cycles ahead, also across layer boundaries);
//   * biases enter as one extra MFMA per tile (A = bias, B = 1 on lane half 0); the sigma and colour heads are
//     VALU dot products over the registers the wave already holds.
// One wave per SIMD (about 400 VGPRs).  Used when nothing has to be saved for backward.
#include "field_common.h"

namespace nerf {

constexpr int RM = 32;  // samples per wave

template <int NFT>
struct WStage {
  float4 w[NFT];
};

template <int KB, int NFT>
__device__ __forceinline__ void stage_load(const float4* __restrict__ seg_lane, int kb, WStage<NFT>& s) {
#pragma unroll
  for (int f = 0; f < NFT; ++f) s.w[f] = seg_lane[(size_t)(f * KB + kb) * 64];
}

// ReLU as ONE instruction: for every non-NaN float max(x, 0) = the float whose bits are max(int bits, 0) (negative floats and
// -0 are negative integers).  fmaxf costs two (hipcc canonicalises its operand first), and next to fp32 MFMAs every VALU
// instruction is matrix time lost (the fp32 MFMA runs on the SIMD's fp32 lanes).
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// acc[f] (+)= sum_k W[f-tile][k] * act(prev)[k]   over KB k-blocks of 8.
// prev = KB/4 register tiles in accumulator layout; RELU_IN: they are the previous layer's raw accumulators and the
// ReLU is applied lazily, one tile at a time, two k-blocks before the tile is first used -- VALU work in the shadow of
// the MFMA stream instead of a serial epilogue between layers.
// Two fragment stages: the A fragments of k-block kb+1 (or k-block 0 of the NEXT segment: NKB k-blocks, NNFT tiles) are
// requested at the top of k-block kb, i.e. 32 MFMAs = 2048 cycles before their first use; the scheduling barrier keeps
// the compiler from sinking the requests towards their uses.  st0 holds k-block 0 on entry and the next segment's
// k-block 0 on exit.  bv (8 bias rows of this lane) != nullptr: one extra MFMA per tile starts the accumulator
// at the bias (A = bias on every lane, B = 1 on lane half 0, C = 0).
// Training (SAVE): the CONSUMER layer writes its activated input tiles to the row-major save buffer (lane (j, h) owns
// the 16-byte groups 32t + 8g + 4h of row j) and, for ReLU inputs, the u16 mask words in the tile kernels' layout.
struct SaveIn {
  float* rows;      // &save[tensor][this lane's row][4h]   (null = nothing to save); lanes past the end of the pass own a dump
                    // row behind the tensor (kernels.h: MSrows), so no store carries a predicate
  uint16_t* mask;   // &masks[layer][tile64][st][h*32 + j] (null = no masks); entry (f, wv) at + (f*2)*256 + wv*64
};

template <int KB, int NFT, int NKB, int NNFT, bool ZERO_INIT, bool RELU_IN, bool SAVE = false>
__device__ __forceinline__ void reg_layer(const int seg, const int next_seg /* float4 offsets into the packed image; < 0: none */, int lane,
                                          const f32x16* prev, f32x16* acc, WStage<8>& st0, const float* bv,
                                          const SaveIn sv, const RegBuf& rb) {
  constexpr int KT = KB / 4;
  WStage<8> st1;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bv != nullptr) {  // accumulators start at the bias (same rounding order as ATen's addmm and as k_field_fwd)
    const float one_h0 = (lane < 32) ? 1.0f : 0.0f;
#pragma unroll
    for (int f = 0; f < NFT; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[f], one_h0, zero, 0, 0, 0);
  }
  f32x16 tin[2];
  auto activate = [&](int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tin[t & 1][r] = RELU_IN ? (SAVE ? fmaxf(prev[t][r], 0.f) : relu1(prev[t][r])) : prev[t][r];  // (the training variant's register allocation falls apart with the integer form)
    // pin the activated tile to this program point; without it the compiler hoists every tile's ReLU to the top of the
    // layer and spills ~370 registers
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(tin[t & 1][r]));
    if (SAVE) {  // compile-time: a SAVE layer always has rows, a SAVE && RELU_IN layer always has masks (no null tests in the stream)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store_row4(sv.rows + 32 * t + 8 * g, make_float4(tin[t & 1][4 * g], tin[t & 1][4 * g + 1], tin[t & 1][4 * g + 2], tin[t & 1][4 * g + 3]));
      if (RELU_IN) {
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= (prev[t][r] > 0.f) ? (1u << r) : 0u;
        sv.mask[((t & 1) * 2) * 256 + (t >> 1) * 64] = (uint16_t)bits;  // feature tile t = (wave t>>1, f = t&1)
      }
    }
  };
  activate(0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    WStage<8>& ld = (kb & 1) ? st0 : st1;
    const WStage<8>& cur = (kb & 1) ? st1 : st0;
    if (kb + 1 < KB) {
#pragma unroll
      for (int f = 0; f < NFT; ++f) ld.w[f] = reg_ldw(rb, seg + (f * KB + kb + 1) * 64);
    } else if (next_seg >= 0) {
#pragma unroll
      for (int f = 0; f < NNFT; ++f) ld.w[f] = reg_ldw(rb, next_seg + (f * NKB) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((kb & 3) == 2 && (kb >> 2) + 1 < KT) activate((kb >> 2) + 1);  // next input tile, behind this k-block's MFMAs
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float b = tin[(kb >> 2) & 1][4 * (kb & 3) + s];
#pragma unroll
      for (int f = 0; f < NFT; ++f) {
        if (ZERO_INIT && bv == nullptr && kb == 0 && s == 0)
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(cur.w[f], s), b, zero, 0, 0, 0);
        else
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(cur.w[f], s), b, acc[f], 0, 0, 0);
      }
    }
#ifdef NERF_FWD_SAVE_SPREAD  // (variant build only -- `make variant NAME=spread DEFS=-DNERF_FWD_SAVE_SPREAD`: the activation's stores one per 8 MFMAs instead of
                             // one burst.  Measured round 5: +10 % on this kernel -- 865 instead of 131 s_nops of hazard padding per tile)
    if (SAVE && NFT == 8 && (kb & 3) == 2 && (kb >> 2) + 1 < KT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);  // the VALU work of a quarter of the tile
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // 1 vector-memory store
      }
    }
#endif
  }
  if (KB & 1) st0 = st1;  // (all segments have an even number of k-blocks: the next k-block 0 already sits in st0)
}

// bias rows of this lane (requested before the layer's k loop, consumed after it)
template <int NFT>
__device__ __forceinline__ void bias_load(const float* __restrict__ bias, int lane, float (&bv)[8]) {
#pragma unroll
  for (int f = 0; f < NFT; ++f) bv[f] = bias[f * 32 + (lane & 31)];
}

#ifdef NERF_STAMPS  // diagnostic build only (make stamps): cycle sums per phase, see scripts/phase_stamps.py
#define RSTAMP(slot)                                            \
  do {                                                          \
    __builtin_amdgcn_sched_barrier(0);                          \
    const unsigned long long t_ = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                          \
    tsum[slot] += t_ - tlast;                                   \
    tlast = t_;                                                 \
  } while (0)
#else
#define RSTAMP(slot) do { } while (0)
#endif

template <bool SAVE, bool DEBUG>
__global__ __launch_bounds__(64, 1) void k_field_fwd_reg(const FieldArgs a) {
#ifdef NERF_STAMPS
  unsigned long long tsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * RM;
  const int m = m0 + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int ray = mc / a.N;
  const float* rf = a.rayf + (size_t)ray * RAYF;
  const float4* wp = a.wp;
  const RegBuf rb = reg_buf(a.wp, threadIdx.x);  // (64-thread workgroups: threadIdx.x is the lane)

  // first fragments of layer 0 are requested before anything else
  WStage<8> st0;
  stage_load<8, 8>(wp + seg_off4(SEG_L0) + lane, 0, st0);

  // ---- sample point and its encoding, straight into B-operand registers:
  // gp[t][4g + s] = gamma_p[k], k = 32t + 8g + 4h + s  (4 consecutive k = two (sin, cos) pairs)
  float p[3];
  sample_point(rf, a.t[mc], p);
  if (DEBUG && a.pts_dbg && valid && h == 0) {
    a.pts_dbg[(size_t)m * 3 + 0] = p[0];
    a.pts_dbg[(size_t)m * 3 + 1] = p[1];
    a.pts_dbg[(size_t)m * 3 + 2] = p[2];
  }
  f32x16 gp[2];
#pragma unroll
  for (int g8 = 0; g8 < 8; ++g8) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pi = 4 * g8 + 2 * h + e;  // (sin, cos) pair index: k = 2 pi
      float sv = 0.f, cv = 0.f;
      if (pi < 30) {
        const int c = pi / 10, l = pi - 10 * c;
        const float x = (c == 0) ? p[0] : ((c == 1) ? p[1] : p[2]);
        const float ph = x * __uint_as_float(kFreqPointBits[l]);
        sincos_phase(ph, sv, cv);
      }
      gp[g8 >> 2][4 * (g8 & 3) + 2 * e] = sv;
      gp[g8 >> 2][4 * (g8 & 3) + 2 * e + 1] = cv;
    }
  }
  if (DEBUG && a.gp_dbg && valid) {
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 8 * g8 + 4 * h + s;
        if (k < POINT_DIM) a.gp_dbg[(size_t)m * POINT_DIM + k] = gp[g8 >> 2][4 * (g8 & 3) + s];
      }
  }

  // training: where this lane's rows / mask words go (rows of the coarse pass first, then the fine pass)
  const size_t MS = (size_t)a.MSrows * WIDTH;
#ifdef NERF_TIMING_SAVE_ALIAS  // (timing experiments only: every save lands in the dump rows -> the stores issue, HBM sees none)
  const long long rrow = a.Mtot + j;
#else
  const long long rrow = valid ? (long long)(a.row0 + m) : a.Mtot + j;  // lanes past the end: dump row
#endif
  float* const srow = SAVE ? a.save + (size_t)rrow * WIDTH + 4 * h : nullptr;
  uint16_t* const mrow = SAVE ? a.masks + ((size_t)(a.tile0 + (m0 >> 6)) * 4 + ((m0 >> 5) & 1)) * 256 + h * 32 + j : nullptr;
  const size_t MKS = (size_t)a.tiles_tot * 4 * 256;
  auto sv_rows = [&](int tensor) { return SaveIn{SAVE ? srow + (size_t)tensor * MS : nullptr, nullptr}; };
  auto sv_relu = [&](int layer) { return SaveIn{SAVE ? srow + (size_t)layer * MS : nullptr, SAVE ? mrow + (size_t)layer * MKS : nullptr}; };

  RSTAMP(0);  // prologue: ray / depth loads, sample point, positional encoding
  // two accumulator sets ping-pong: a layer reads the previous layer's raw accumulators (ReLU applied lazily)
  f32x16 A[8], B[8];
  constexpr int L256 = 8 * 32 * 64;  // float4 per 256x256 segment
  constexpr int sL1 = seg_off4(SEG_L1);
  constexpr int sL5 = seg_off4(SEG_L5);
  float bv[8];

  // ---- layer 0: gamma_p 60(64) -> 256
  bias_load<8>(a.w.p[B_L0], lane, bv);
  reg_layer<8, 8, 32, 8, true, false, SAVE>(seg_off4(SEG_L0), sL1, lane, gp, A, st0, bv, sv_rows(S_GP), rb);
  RSTAMP(1);  // layer 0 (264 MFMAs)
  // ---- layers 1..3 (the segment after L3 is L4A: same shape)
  bias_load<8>(a.w.p[3], lane, bv);
  reg_layer<32, 8, 32, 8, true, true, SAVE>(sL1, sL1 + L256, lane, A, B, st0, bv, sv_relu(0), rb);
  bias_load<8>(a.w.p[5], lane, bv);
  reg_layer<32, 8, 32, 8, true, true, SAVE>(sL1 + L256, sL1 + 2 * L256, lane, B, A, st0, bv, sv_relu(1), rb);
  bias_load<8>(a.w.p[7], lane, bv);
  reg_layer<32, 8, 32, 8, true, true, SAVE>(sL1 + 2 * L256, seg_off4(SEG_L4A), lane, A, B, st0, bv, sv_relu(2), rb);
  RSTAMP(2);  // layers 1..3 (3,096 MFMAs)
  // ---- layer 4: cat(h3, gamma_p), hidden first (nerf.py:109)
  bias_load<8>(a.w.p[9], lane, bv);
  reg_layer<32, 8, 8, 8, true, true, SAVE>(seg_off4(SEG_L4A), seg_off4(SEG_L4B), lane, B, A, st0, bv, sv_relu(3), rb);
  reg_layer<8, 8, 32, 8, false, false>(seg_off4(SEG_L4B), sL5, lane, gp, A, st0, nullptr, SaveIn{nullptr, nullptr}, rb);
  RSTAMP(3);  // layer 4 (1,288 MFMAs)
  // ---- layers 5..7 (the segment after L7 is the folded point_info / dir_info layer: 4 tiles)
  bias_load<8>(a.w.p[11], lane, bv);
  reg_layer<32, 8, 32, 8, true, true, SAVE>(sL5, sL5 + L256, lane, A, B, st0, bv, sv_relu(4), rb);
  bias_load<8>(a.w.p[13], lane, bv);
  reg_layer<32, 8, 32, 8, true, true, SAVE>(sL5 + L256, sL5 + 2 * L256, lane, B, A, st0, bv, sv_relu(5), rb);
  bias_load<8>(a.w.p[15], lane, bv);
  reg_layer<32, 8, 32, 4, true, true, SAVE>(sL5 + 2 * L256, seg_off4(SEG_FOLD), lane, A, B, st0, bv, sv_relu(6), rb);
  RSTAMP(4);  // layers 5..7 (3,096 MFMAs)
  // ---- sigma head on h7 = relu(B) (VALU): sigma = |w_sigma . h7 + b|  (nerf.py:94, 115)
  {
    const float* ws = a.w.p[W_SIGMA] + 4 * h;
    float s = 0.f;
    // weights one tile ahead of their use, in explicit groups: under the register pressure of the training variant the
    // compiler otherwise issues the 32 loads one at a time, each followed by a full wait
    float4 wq[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wq[0][g] = *reinterpret_cast<const float4*>(ws + 8 * g);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t + 1 < 8) {
#pragma unroll
        for (int g = 0; g < 4; ++g) wq[(t + 1) & 1][g] = *reinterpret_cast<const float4*>(ws + 32 * (t + 1) + 8 * g);
      }
      if (SAVE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q = wq[t & 1][g];
        s = __builtin_fmaf(relu1(B[t][4 * g + 0]), q.x, s);
        s = __builtin_fmaf(relu1(B[t][4 * g + 1]), q.y, s);
        s = __builtin_fmaf(relu1(B[t][4 * g + 2]), q.z, s);
        s = __builtin_fmaf(relu1(B[t][4 * g + 3]), q.w, s);
      }
      if (SAVE) __builtin_amdgcn_sched_barrier(0);
    }
    s += __shfl_xor(s, 32);
    if (valid && h == 0) {
      const float pre = s + a.w.p[B_SIGMA][0];
      a.sigma[m] = fabsf(pre);
      if (SAVE) a.spre[a.row0 + m] = pre;
    }
  }
  RSTAMP(5);  // sigma head
  // ---- point_info (256 -> 256, no activation) and dir_info's feature columns as ONE 128 x 256 layer on h7 (common.h SEG_FOLD):
  // c = relu(W_fold h7 + dvec), dvec (per ray) = b_dir + W_dir[:, :24] gamma_d + W_dir[:, 24:] b_pi = the accumulator start
  {
    const float* dv = a.dvec + (size_t)ray * HALF + 4 * h;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(dv + 32 * f + 8 * g);
        A[f][4 * g + 0] = q.x;
        A[f][4 * g + 1] = q.y;
        A[f][4 * g + 2] = q.z;
        A[f][4 * g + 3] = q.w;
      }
  }
  reg_layer<32, 4, 32, 4, false, true, SAVE>(seg_off4(SEG_FOLD), -1, lane, B, A, st0, nullptr, sv_relu(7), rb);
  RSTAMP(6);  // point_info + dir_info folded (512 MFMAs)
  // ---- colour head (VALU): rgb = sigmoid(W_c relu(.) + b)  (nerf.py:99, 119)
  {
    const float* wc = a.w.p[W_COLOR] + 4 * h;
    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q0 = *reinterpret_cast<const float4*>(wc + 32 * t + 8 * g);
        const float4 q1 = *reinterpret_cast<const float4*>(wc + HALF + 32 * t + 8 * g);
        const float4 q2 = *reinterpret_cast<const float4*>(wc + 2 * HALF + 32 * t + 8 * g);
        const float c0 = relu1(A[t][4 * g + 0]), c1 = relu1(A[t][4 * g + 1]);
        const float c2 = relu1(A[t][4 * g + 2]), c3 = relu1(A[t][4 * g + 3]);
        if (SAVE) store_row4(srow + S_C * MS + 32 * t + 8 * g, make_float4(c0, c1, c2, c3));
        z0 = __builtin_fmaf(c3, q0.w, __builtin_fmaf(c2, q0.z, __builtin_fmaf(c1, q0.y, __builtin_fmaf(c0, q0.x, z0))));
        z1 = __builtin_fmaf(c3, q1.w, __builtin_fmaf(c2, q1.z, __builtin_fmaf(c1, q1.y, __builtin_fmaf(c0, q1.x, z1))));
        z2 = __builtin_fmaf(c3, q2.w, __builtin_fmaf(c2, q2.z, __builtin_fmaf(c1, q2.y, __builtin_fmaf(c0, q2.x, z2))));
      }
    z0 += __shfl_xor(z0, 32);
    z1 += __shfl_xor(z1, 32);
    z2 += __shfl_xor(z2, 32);
    if (valid && h == 0) {
      const float* bc = a.w.p[B_COLOR];
      a.rgb[(size_t)m * 3 + 0] = 1.0f / (1.0f + expf(-(z0 + bc[0])));
      a.rgb[(size_t)m * 3 + 1] = 1.0f / (1.0f + expf(-(z1 + bc[1])));
      a.rgb[(size_t)m * 3 + 2] = 1.0f / (1.0f + expf(-(z2 + bc[2])));
    }
  }
#ifdef NERF_STAMPS
  RSTAMP(7);  // colour head + stores
  if (a.stamps && lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, tsum[i]);
    atomicAdd(a.stamps + 31, 1ull);
  }
#endif
}

hipError_t launch_field_fwd_reg(const FieldArgs& a, bool save, hipStream_t st) {
  const int tiles = (a.M + RM - 1) / RM;
  if (save)
    hipLaunchKernelGGL((k_field_fwd_reg<true, false>), dim3(tiles), dim3(64), 0, st, a);
  else if (a.pts_dbg || a.gp_dbg)
    hipLaunchKernelGGL((k_field_fwd_reg<false, true>), dim3(tiles), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL((k_field_fwd_reg<false, false>), dim3(tiles), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace nerf
