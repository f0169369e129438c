// field_fwd_bf16.hip -- field query with a bf16 MLP (v_mfma_f32_32x32x16_bf16) for MI355X / gfx950.
//
// Same structure idea as field_fwd_reg.hip -- a wave owns 32 samples and all 256 features of every layer, activations
// never leave its registers -- re-balanced for a matrix pipe that is 16x faster:
//   * at 32 cycles per MFMA a wave needs 1 KiB of weights every 32 cycles; four waves per CU streaming that from L2
//     would need 128 B/clk/CU, twice what the vector memory path delivers.  So the weight stream (bf16_common.h) goes
//     through LDS: the 8 waves of a workgroup (2 per SIMD, 256 samples) consume the SAME fragment sequence, which is
//     staged chunk by chunk (16 KiB = 16 fragments) into an 8-slot LDS ring by direct-to-LDS loads (global_load_lds_dwordx4,
//     two per wave and chunk), six chunks ahead of use; every fragment is then one conflict-free ds_read_b128;
//   * one s_barrier per chunk, placed in the MIDDLE of the previous chunk: it publishes chunk c+1 (each wave first waits,
//     with a counted vmcnt, for its own two pieces) and frees the slot of chunk c-1 for the next load, while the fragment
//     reads and MFMAs of chunk c continue on both sides of it;
//   * output-tile-major order: one 16-register accumulator at a time runs over all k-steps of its tile, then is
//     bias-free ReLU'd, rounded to bf16 pairwise (v_cvt_pk_bf16_f32 + v_pk_max_i16) and IS the next layer's B operand;
//     a layer's input and output live as 64 + 64 packed registers, so two waves fit on a SIMD and one wave's VALU work
//     (encoding, conversions, heads) runs in the shadow of the other's MFMAs;
//   * biases are fp32 accumulator start values read from LDS; sigma and colour heads are extra MFMA tiles (row 0 / rows
//     0..2 of a 32-row tile), 24 MFMAs instead of ~400 VALU instructions.
#include "bf16_common.h"
#include "field_common.h"

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BF_NS = 8;        // LDS ring slots
constexpr int BF_SYNC_POS = 8;  // fragment position inside a chunk at which the next chunk is published
constexpr int BF_D = 6;         // fragment reads in flight per wave (<= BF_CHUNK - BF_SYNC_POS)
constexpr int BF_WG = 512;      // 8 waves x 32 samples
constexpr int BF_LDS_BYTES = BF_BIAS_BYTES + BF_NS * BF_CHUNK * BF_FRAG_BYTES;
static_assert(BF_D <= BF_CHUNK - BF_SYNC_POS, "a prefetched fragment must not lie in an unpublished chunk");

__device__ __forceinline__ unsigned pack2(float a, float b) {  // two fp32 -> two bf16 (RNE), a in the low half
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned pack2_relu(float a, float b) {
  const f32x2 v = {a, b};
  s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(v, bf16x2));
  const s16x2 z = {0, 0};
  s = __builtin_elementwise_max(s, z);  // bf16 as int16: negative values (sign bit) -> 0, positive order preserved
  return __builtin_bit_cast(unsigned, s);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {  // n is a constant after unrolling
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 12: wait_vmcnt<12>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

struct BfCtx {
  const unsigned char* wimg;  // global: bias block + fragment stream
  unsigned char* lds;         // bias block + ring
  unsigned lds_base;          // the same as an LDS byte address
  int lane, wv;
};

typedef void __attribute__((address_space(3)))* lptr_t;

// One direct-to-LDS load: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane*16 (lds_dst wave-uniform).
// Inline asm on purpose: hipcc treats a builtin LDS-DMA as a pending write to the whole LDS array and drains the load
// queue (vmcnt(0)) in front of unrelated ds_reads; hidden from it, the loads are ordered by bf_sync's counted waits alone.
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// this wave's two 1-KiB pieces of chunk c -> ring slot c % BF_NS
__device__ __forceinline__ void bf_dma_chunk(const BfCtx& c, int chunk) {
  const int slot = chunk % BF_NS;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int fr = 2 * c.wv + e;
    glds16(c.wimg + BF_BIAS_BYTES + ((size_t)chunk * BF_CHUNK + fr) * BF_FRAG_BYTES + c.lane * 16,
           c.lds_base + BF_BIAS_BYTES + (slot * BF_CHUNK + fr) * BF_FRAG_BYTES);
  }
}

// number of this wave's loads that may stay in flight when chunk c + 1 must have landed
__device__ __forceinline__ constexpr int bf_inflight_after(int c) {
  const int last = (c + BF_NS - 2 < BF_NCHUNK - 1) ? c + BF_NS - 2 : BF_NCHUNK - 1;  // newest chunk requested so far
  return (last >= c + 2) ? 2 * (last - (c + 2) + 1) : 0;
}

// executed by every wave at fragment position BF_SYNC_POS of chunk c
__device__ __forceinline__ void bf_sync(const BfCtx& c, int chunk) {
  wait_vmcnt_dyn(bf_inflight_after(chunk));  // my pieces of chunk + 1 are in LDS ...
  __builtin_amdgcn_s_barrier();              // ... and so are everybody's; everybody is past chunk - 1
  asm volatile("" ::: "memory");             // no LDS read may be moved above the barrier by the compiler
  if (chunk + BF_NS - 1 < BF_NCHUNK) bf_dma_chunk(c, chunk + BF_NS - 1);  // into the slot of chunk - 1
}

__device__ __forceinline__ u32x4 bf_frag(const BfCtx& c, int idx) {
  const int slot = (idx / BF_CHUNK) % BF_NS;
  return *reinterpret_cast<const u32x4*>(c.lds + BF_BIAS_BYTES + (slot * BF_CHUNK + idx % BF_CHUNK) * BF_FRAG_BYTES + c.lane * 16);
}

__device__ __forceinline__ f32x16 bf_bias_tile(const BfCtx& c, int tile) {
  const int h = c.lane >> 5;
  f32x16 a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = *reinterpret_cast<const float4*>(c.lds + (tile * 32 + 8 * g + 4 * h) * 4);
    a[4 * g + 0] = q.x;
    a[4 * g + 1] = q.y;
    a[4 * g + 2] = q.z;
    a[4 * g + 3] = q.w;
  }
  return a;
}

__device__ __forceinline__ f32x16 bf_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One segment of the stream: NFT output tiles x (KSA + KSB) k-steps starting at fragment S0; inputs inA (k-steps
// 0..KSA-1) then inB.  fr = ring of BF_D prefetched fragments (fr[idx % BF_D] holds fragment idx on entry to step idx).
// Two accumulators alternate so that nothing waits for an MFMA result: tile f runs in acc[(P0 + f) & 1]; the finished
// accumulator of tile f-1 is consumed by epi(f-1, .) BF_EPI_POS k-steps into tile f (the last tile of the previous
// segment by prev_epi), and right after that it is re-started at the bias of tile f+1 (or of the next segment's tile 0,
// bias tile NEXT_BT), an LDS read with >= 1 k-steps of MFMAs to land in.
constexpr int BF_EPI_POS = 2;
template <int S0, int NFT, int KSA, int KSB, int BT0, int P0, int NEXT_BT, class Epi, class PrevEpi>
__device__ __forceinline__ void bf_segment(const BfCtx& c, u32x4 (&fr)[BF_D], f32x16 (&acc)[2], const u32x4* inA, const u32x4* inB,
                                           Epi&& epi, PrevEpi&& prev_epi) {
  constexpr int KS = KSA + KSB;
  static_assert(KS > BF_EPI_POS + 1, "segment too short for the deferred epilogue");
#pragma unroll
  for (int f = 0; f < NFT; ++f) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int idx = S0 + f * KS + ks;
      if (idx % BF_CHUNK == BF_SYNC_POS) bf_sync(c, idx / BF_CHUNK);
      const u32x4 a = fr[idx % BF_D];
      if (idx + BF_D < BF_NFRAG) fr[idx % BF_D] = bf_frag(c, idx + BF_D);
      acc[(P0 + f) & 1] = bf_mfma(a, ks < KSA ? inA[ks] : inB[ks - KSA], acc[(P0 + f) & 1]);
      if (ks == BF_EPI_POS) {
        if (f == 0)
          prev_epi(acc[(P0 + 1) & 1]);
        else
          epi(f - 1, acc[(P0 + f + 1) & 1]);
        if (f + 1 < NFT)
          acc[(P0 + f + 1) & 1] = bf_bias_tile(c, BT0 + f + 1);
        else if (NEXT_BT >= 0)
          acc[(P0 + f + 1) & 1] = bf_bias_tile(c, NEXT_BT);
      }
    }
  }
}

template <bool SAVE>
__global__ __launch_bounds__(BF_WG, 1) void k_field_fwd_bf16(const FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = c.lane, j = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * (BF_WG / 2) + c.wv * 32 + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int ray = mc / a.N;

  // ---- ordinary loads first (the compiler drains them with vmcnt(0) before first use; nothing else is in flight yet)
  const float* rf = a.rayf + (size_t)ray * RAYF;
  float p[3], dw[3];
  sample_point(rf, a.t[mc], p);
#pragma unroll
  for (int i = 0; i < 3; ++i) dw[i] = rf[RF_DWRD + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(p[i]), "+v"(dw[i]));  // values are in registers from here on

  // ---- start the weight stream: bias block (as "chunk -1": 2 pieces per wave) and chunks 0 .. BF_NS-2
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int fr = 2 * c.wv + e;
    glds16(c.wimg + fr * BF_FRAG_BYTES + lane * 16, c.lds_base + fr * BF_FRAG_BYTES);
  }
#pragma unroll
  for (int ch = 0; ch < BF_NS - 1; ++ch) bf_dma_chunk(c, ch);

  // ---- positional encodings straight into B-operand registers (fp32 values as in the fp32 path, rounded to bf16):
  // k-step ks, slot pair (s, s+1): features k = 16ks + 4h + {0,1 | 2,3 | 8,9 | 10,11} = (sin, cos) pairs
  // pi = 8ks + 2h + {0, 1, 4, 5}
  u32x4 gp[4], gd[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pi = 8 * ks + 2 * h + (q & 1) + 4 * (q >> 1);
      float sv = 0.f, cv = 0.f;
      if (pi < 30) {
        const int cc = pi / 10, l = pi - 10 * cc;
        const float x = (cc == 0) ? p[0] : ((cc == 1) ? p[1] : p[2]);
        sincos_phase(x * __uint_as_float(kFreqPointBits[l]), sv, cv);
      }
      gp[ks][q] = pack2(sv, cv);
    }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pi = 8 * ks + 2 * h + (q & 1) + 4 * (q >> 1);
      float sv = 0.f, cv = 0.f;
      if (pi < 12) {
        const int cc = pi / 4, l = pi - 4 * cc;
        const float x = (cc == 0) ? dw[0] : ((cc == 1) ? dw[1] : dw[2]);
        sincos_phase(x * __uint_as_float(kFreqDirBits[l]), sv, cv);
      }
      gd[ks][q] = pack2(sv, cv);
    }

  // ---- bias block and chunk 0 have landed (mine), then everybody's
  wait_vmcnt<2 * (BF_NS - 2)>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32x4 fr[BF_D];
#pragma unroll
  for (int i = 0; i < BF_D; ++i) fr[i] = bf_frag(c, i);

  u32x4 X[16], Y[16];
  f32x16 acc[2];
  acc[0] = bf_bias_tile(c, BFB_L0);
  auto relu_to = [&](u32x4* out) {
    return [out](int f, const f32x16& A) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[2 * f + mh][q] = pack2_relu(A[8 * mh + 2 * q], A[8 * mh + 2 * q + 1]);
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](const f32x16& A) { epi(f, A); }; };
  auto nothing = [](const f32x16&) {};
  auto nothing_f = [](int, const f32x16&) {};

  // ---- layers 0..7 (nerf.py:104-112)
  bf_segment<BFS_L0, 8, 4, 0, BFB_L0, 0, BFB_L0 + 8>(c, fr, acc, gp, nullptr, relu_to(X), nothing);
  bf_segment<BFS_L1, 8, 16, 0, BFB_L0 + 8, 0, BFB_L0 + 16>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 7));
  bf_segment<BFS_L1 + 128, 8, 16, 0, BFB_L0 + 16, 0, BFB_L0 + 24>(c, fr, acc, Y, nullptr, relu_to(X), last_of(relu_to(Y), 7));
  bf_segment<BFS_L1 + 256, 8, 16, 0, BFB_L0 + 24, 0, BFB_L0 + 32>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 7));
  bf_segment<BFS_L4, 8, 16, 4, BFB_L0 + 32, 0, BFB_L0 + 40>(c, fr, acc, Y, gp, relu_to(X), last_of(relu_to(Y), 7));
  bf_segment<BFS_L5, 8, 16, 0, BFB_L0 + 40, 0, BFB_L0 + 48>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 7));
  bf_segment<BFS_L5 + 128, 8, 16, 0, BFB_L0 + 48, 0, BFB_L0 + 56>(c, fr, acc, Y, nullptr, relu_to(X), last_of(relu_to(Y), 7));
  bf_segment<BFS_L5 + 256, 8, 16, 0, BFB_L0 + 56, 0, BFB_PI>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 7));
  // ---- point_info (no activation) + sigma head (tile 8, row 0): sigma = |w_sigma . h7 + b|  (nerf.py:94, 113-115)
  float spre = 0.f;
  auto pi_epi = [&](int f, const f32x16& A) {
    if (f < 8) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[2 * f + mh][q] = pack2(A[8 * mh + 2 * q], A[8 * mh + 2 * q + 1]);
    } else {
      spre = A[0];
    }
  };
  bf_segment<BFS_PI, 9, 16, 0, BFB_PI, 0, BFB_DIR>(c, fr, acc, Y, nullptr, pi_epi, last_of(relu_to(Y), 7));
  // ---- dir_info on cat(gamma_d, feat), ReLU (nerf.py:117-118); its first tile also retires the sigma tile
  bf_segment<BFS_DIR, 4, 2, 16, BFB_DIR, 1, BFB_COL>(c, fr, acc, gd, X, relu_to(Y), last_of(pi_epi, 8));
  if (valid && h == 0) {
    a.sigma[m] = fabsf(spre);
    if (SAVE) a.spre[a.row0 + m] = spre;
  }
  // ---- colour head: rows 0..2 of one tile, sigmoid (nerf.py:99, 119)
  bf_segment<BFS_COL, 1, 8, 0, BFB_COL, 1, -1>(c, fr, acc, Y, nullptr, nothing_f, last_of(relu_to(Y), 3));
  if (valid && h == 0) {
    a.rgb[(size_t)m * 3 + 0] = 1.0f / (1.0f + expf(-acc[1][0]));
    a.rgb[(size_t)m * 3 + 1] = 1.0f / (1.0f + expf(-acc[1][1]));
    a.rgb[(size_t)m * 3 + 2] = 1.0f / (1.0f + expf(-acc[1][2]));
  }
}

// ------------------------------------------------------------------------------------------
// weight image
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf_weight(const Weights24& w, int frag, int i, int kk /* 0..15 inside the k-step */, int h) {
  (void)h;
  if (frag < BFS_L1) {  // L0
    const int f = frag / 4, ks = frag % 4, k = 16 * ks + kk;
    return k < POINT_DIM ? w.p[0][(size_t)(32 * f + i) * POINT_DIM + k] : 0.f;
  }
  if (frag < BFS_L4) {  // L1..L3
    const int r = frag - BFS_L1, l = 1 + r / 128, q = r % 128, f = q / 16, ks = q % 16;
    return w.p[2 * l][(size_t)(32 * f + i) * WIDTH + 16 * ks + kk];
  }
  if (frag < BFS_L5) {  // L4: [256][316] = cat(hidden, gamma_p)
    const int q = frag - BFS_L4, f = q / 20, ks = q % 20, k = 16 * ks + kk;
    return (k < WIDTH + POINT_DIM) ? w.p[8][(size_t)(32 * f + i) * (WIDTH + POINT_DIM) + k] : 0.f;
  }
  if (frag < BFS_PI) {  // L5..L7
    const int r = frag - BFS_L5, l = 5 + r / 128, q = r % 128, f = q / 16, ks = q % 16;
    return w.p[2 * l][(size_t)(32 * f + i) * WIDTH + 16 * ks + kk];
  }
  if (frag < BFS_DIR) {  // point_info + sigma row
    const int q = frag - BFS_PI, f = q / 16, ks = q % 16, k = 16 * ks + kk;
    if (f < 8) return w.p[W_PI][(size_t)(32 * f + i) * WIDTH + k];
    return i == 0 ? w.p[W_SIGMA][k] : 0.f;
  }
  if (frag < BFS_COL) {  // dir_info: [128][280] = cat(gamma_d (24), feat)
    const int q = frag - BFS_DIR, f = q / 18, ks = q % 18, k = 16 * ks + kk;
    if (ks < 2) return k < DIR_DIM ? w.p[W_DIR][(size_t)(32 * f + i) * (WIDTH + DIR_DIM) + k] : 0.f;
    return w.p[W_DIR][(size_t)(32 * f + i) * (WIDTH + DIR_DIM) + DIR_DIM + (k - 32)];
  }
  const int ks = frag - BFS_COL, k = 16 * ks + kk;  // colour head
  return i < 3 ? w.p[W_COLOR][(size_t)i * HALF + k] : 0.f;
}

__device__ __forceinline__ float bf_bias(const Weights24& w, int tile, int i) {
  if (tile < BFB_PI) return w.p[2 * (tile / 8) + 1][32 * (tile % 8) + i];
  if (tile < BFB_SIGMA) return w.p[B_PI][32 * (tile - BFB_PI) + i];
  if (tile == BFB_SIGMA) return i == 0 ? w.p[B_SIGMA][0] : 0.f;
  if (tile < BFB_COL) return w.p[B_DIR][32 * (tile - BFB_DIR) + i];
  return i < 3 ? w.p[B_COLOR][i] : 0.f;
}

__global__ __launch_bounds__(256) void k_pack_weights_bf16(const Weights24 w, unsigned char* __restrict__ img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid < BF_NFRAG * 64) {
    const int frag = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
      v[q] = pack2(bf_weight(w, frag, i, kk, h), bf_weight(w, frag, i, kk + 1, h));
    }
    *reinterpret_cast<u32x4*>(img + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
  } else {
    const int b = gid - BF_NFRAG * 64;
    if (b < BF_BIAS_BYTES / 4) {
      const int tile = b >> 5, i = b & 31;
      reinterpret_cast<float*>(img)[b] = tile < BF_NBIAS_TILES ? bf_bias(w, tile, i) : 0.f;
    }
  }
}

hipError_t launch_pack_weights_bf16(const Weights24& w, unsigned char* img, hipStream_t st) {
  const int threads = BF_NFRAG * 64 + BF_BIAS_BYTES / 4;
  hipLaunchKernelGGL(k_pack_weights_bf16, dim3((threads + 255) / 256), dim3(256), 0, st, w, img);
  return hipGetLastError();
}

hipError_t launch_field_fwd_bf16(const FieldArgs& a, bool save, hipStream_t st) {
  static bool attr_done = false;  // >64 KiB of dynamic LDS needs an opt-in, once per process and device function
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_fwd_bf16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BF_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  (void)save;
  const int wgs = (a.M + BF_WG / 2 - 1) / (BF_WG / 2);
  hipLaunchKernelGGL((k_field_fwd_bf16<false>), dim3(wgs), dim3(BF_WG), BF_LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace nerf
