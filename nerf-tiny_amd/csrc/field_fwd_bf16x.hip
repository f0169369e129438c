// field_fwd_bf16x.hip -- INFERENCE forward of the bf16-MLP field query on v_mfma_f32_16x16x32_bf16 (MI355X / gfx950).
//
// Same machinery and same arithmetic specification as field_fwd_bf16.hip (bf16_stream.h: LDS ring of 1-KiB weight
// fragments, 8 waves x 32 samples, activations in registers, accumulator = next layer's operand), built on the 16x16x32
// form of the bf16 MFMA: under an MFMA load this chip is power-limited, and on random operands the 16x16x32 form sustains
// 2.09 PFLOP/s where the 32x32x16 form sustains 1.7-1.8 (scripts/micro/mfma_bf16_shapes.hip) -- same cycles per FLOP, a
// higher clock.  What changes:
//   * a fragment is 16 output features x 32 inputs (still 1 KiB, still one per 32 cycles of MFMA: it feeds TWO MFMAs, one
//     per 16-sample half of the wave's 32 samples);
//   * lane (n, q) = (lane & 15, lane >> 4) holds, for its two samples n and 16 + n, features 4q..4q+3 of every 16-feature
//     accumulator tile (4 registers per tile and half); tiles 2s and 2s+1, ReLU'd and rounded pairwise, are the 8 k-slots
//     of that lane for k-step s of the next layer -- slot j <-> input feature 32s + 16 (j >> 2) + 4q + (j & 3), which is the
//     order the weight image stores;
//   * biases: one 16-byte LDS read per tile (shared by both halves); sigma / colour heads: row 0 / rows 0..2 of an extra
//     16-row tile, read from the lanes with q == 0.
// Training keeps the 32x32x16 kernels (their save layout is what the chain and dW kernels read); the two forms round the
// same bf16 operands but sum in a different order, so an inference and a training forward of the same batch agree to
// fp32 rounding before the next bf16 rounding, not bit for bit.
#include "bf16_stream.h"
#include "bf16_weights.h"
#include "ray_parts.h"
#include "prep_parts.h"

namespace nerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// bias block: the float layout of the 32x32x16 image (bf16_common.h), addressed per 16 features
constexpr int BXB_SIGMA = 32 * BFB_SIGMA, BXB_DIR = 32 * BFB_DIR, BXB_COL = 32 * BFB_COL;

constexpr int BX_SMALL_MAX_WGS = 256;  // 4-wave workgroups up to this many of them (one per CU)
constexpr int BX_LDS_BYTES = BF_LDS_BYTES + BF_WG * 32;  // bias block + ring + 32 bytes per lane of parked encodings = 160 KiB
static_assert(BX_LDS_BYTES <= 160 * 1024, "LDS of one CU");

struct BxStream {
  static constexpr int NFRAG = BX_NFRAG, NCHUNK = BX_NCHUNK, NS = BF_NS, RING_OFF = BF_BIAS_BYTES, D = BF_D;
  static constexpr bool HAS_BIAS = true;
  static constexpr int PROLOGUE_STORES = 0;
  __device__ static constexpr int stores_before(int) { return 0; }
};
// The two-column form (NH = 4): ONE wave per SIMD holds 64 samples (four 16-sample groups, 512 registers), 4 waves = the same 256
// samples per workgroup and per pass over the weight image, but every fragment read from LDS feeds FOUR MFMAs instead of two.
struct BxStream4 : BxStream {
  static constexpr int PW = 4;  // 16 pieces of a chunk over 4 waves
};
// A 4-wave workgroup of the shipped form (NH = 2, WAVES = 4: 128 samples per workgroup) is what a SMALL pass gets: 32,768 coarse
// samples (512 rays) are 128 workgroups of 256 samples -- half of the CUs idle -- or 256 of 128.
template <int WAVES> struct BxStreamOf { using type = BxStream; };
template <> struct BxStreamOf<4> { using type = BxStream4; };

template <int NH> struct AccN { f32x4 c[NH]; };  // one 16-feature tile for the wave's NH 16-sample groups

__device__ __forceinline__ f32x4 bx_mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// NT output tiles x (KSA + KSB) k-steps starting at fragment S0; inputs inA[half][k-step] then inB.  Tile f runs in
// acc[(P0 + f) & 1]; the finished accumulators of tile f-1 are consumed by epi(f-1, .) right after the second k-step of tile
// f (prev_epi: the last tile of the previous segment) and re-started at the bias of tile f+1 (bias float offset B0 + 16 f;
// NEXT_B: of the next segment's tile 0, < 0: none).
template <class S, int NH, int S0, int NT, int KSA, int KSB, int B0, int P0, int NEXT_B, class Epi, class PrevEpi>
__device__ __forceinline__ void bx_segment(const BfCtx& c, u32x4 (&fr)[BF_D], AccN<NH> (&acc)[2], const u32x4 (*inA)[8], const u32x4 (*inB)[8],
                                           Epi&& epi, PrevEpi&& prev_epi) {
  constexpr int KS = KSA + KSB;
  static_assert(KS >= 2, "segment too short for the deferred epilogue");
  const int q = c.lane >> 4;
  auto bias = [&](int off) {
    const float4 v = *reinterpret_cast<const float4*>(c.lds + (off + 4 * q) * 4);
    AccN<NH> a;
#pragma unroll
    for (int h = 0; h < NH; ++h) a.c[h] = f32x4{v.x, v.y, v.z, v.w};
    return a;
  };
  static_for<NT * KS>([&](auto I) {
    constexpr int f = I / KS, ks = I % KS, idx = S0 + I;
    constexpr int cur = (P0 + f) & 1, oth = (P0 + f + 1) & 1;
    if constexpr (idx % BF_CHUNK == BF_SYNC_POS) bf_sync<S, idx / BF_CHUNK>(c);
    const u32x4 a = fr[idx % S::D];
    if constexpr (idx + S::D < S::NFRAG) fr[idx % S::D] = bf_frag<S>(c, idx + S::D);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if constexpr (ks < KSA)
        acc[cur].c[h] = bx_mfma(a, inA[h][ks], acc[cur].c[h]);
      else
        acc[cur].c[h] = bx_mfma(a, inB[h][ks - KSA], acc[cur].c[h]);
    }
    if constexpr (ks == 1) {  // four MFMAs (64 cycles) after the previous tile's last one
      if constexpr (f == 0)
        prev_epi(acc[oth]);
      else
        epi(f - 1, acc[oth]);
      if constexpr (f + 1 < NT)
        acc[oth] = bias(B0 + 16 * (f + 1));
      else if constexpr (NEXT_B >= 0)
        acc[oth] = bias(NEXT_B);
    }
  });
}

// One pass of the field MLP over the wave's 16 NH samples, the weight stream S running through the workgroup's LDS ring once: inputs =
// this lane's sample points p[h] and world directions dw[h]; `mid(spre)` is called between the dir_info and the colour segments with the
// sigma pre-activations (valid on the lanes with q == 0: lane n = sample n of group h); on return cpre[h][0..2] are the colour head's
// pre-activations (same lanes).  Every wave of the workgroup must call it (one barrier per chunk).  S::HAS_BIAS = false: the bias block is
// already in LDS (a second pass of the same kernel).
template <class S, int NH, class Mid>
__device__ __forceinline__ void bx_field_pass(const BfCtx& c, unsigned char* lds, const float (&p)[NH][3], const float (&dw)[NH][3], Mid&& mid,
                                              float (&cpre)[NH][3]) {
  using Acc = AccN<NH>;
  const int q = c.lane >> 4;
  bf_stream_start<S>(c);

  // ---- encodings straight into B-operand registers: k-step s, slot pair (j, j+1) = (sin, cos) pair
  // pi = 16s + 8 (j >> 2) + 2q + ((j >> 1) & 1), i.e. features 32s + 16 (j >> 2) + 4q + (j & 3)
  u32x4 gp[NH][8], gd[NH][8];  // only [.][0..1] / [.][0] are used (the segment interface indexes [group][k-step])
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pi = 16 * s + 8 * (e >> 1) + 2 * q + (e & 1);
        float sv = 0.f, cv = 0.f;
        if (pi < 30) {
          const int cc = pi / 10, l = pi - 10 * cc;
          const float x = (cc == 0) ? p[h][0] : ((cc == 1) ? p[h][1] : p[h][2]);
          sincos_phase(x * __uint_as_float(kFreqPointBits[l]), sv, cv);
        }
        gp[h][s][e] = pack2(sv, cv);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pi = 8 * (e >> 1) + 2 * q + (e & 1);
      float sv = 0.f, cv = 0.f;
      if (pi < 12) {
        const int cc = pi / 4, l = pi - 4 * cc;
        const float x = (cc == 0) ? dw[h][0] : ((cc == 1) ? dw[h][1] : dw[h][2]);
        sincos_phase(x * __uint_as_float(kFreqDirBits[l]), sv, cv);
      }
      gd[h][0][e] = pack2(sv, cv);
    }
  }

  // the direction encodings are needed 1,100 fragments later: parked in the 16 KiB of LDS behind the ring (32 bytes per lane)
  // instead of 8 registers the allocator would spill to scratch
  u32x4* const gd_park = reinterpret_cast<u32x4*>(lds + BF_LDS_BYTES) + NH * (c.wv * 64 + c.lane);
#pragma unroll
  for (int h = 0; h < NH; ++h) gd_park[h] = gd[h][0];

  u32x4 fr[BF_D];
  bf_stream_first<S>(c, fr);

  u32x4 X[NH][8], Y[NH][8];
  Acc acc[2];
  {
    const float4 v = *reinterpret_cast<const float4*>(lds + (4 * q) * 4);
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[0].c[h] = f32x4{v.x, v.y, v.z, v.w};
  }
  // epilogue of a ReLU layer: tile f (16 features) -> slots 2 (f & 1), 2 (f & 1) + 1 of k-step f >> 1 of the next layer
  auto relu_to = [&](u32x4 (*out)[8]) {
    return [out](int f, const Acc& A) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        out[h][f >> 1][2 * (f & 1) + 0] = pack2_relu(A.c[h][0], A.c[h][1]);
        out[h][f >> 1][2 * (f & 1) + 1] = pack2_relu(A.c[h][2], A.c[h][3]);
      }
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](const Acc& A) { epi(f, A); }; };
  auto nothing = [](const Acc&) {};
  auto nothing_f = [](int, const Acc&) {};

  // ---- layers 0..7 (nerf.py:104-112)
  bx_segment<S, NH, BXS_L0, 16, 2, 0, 0 * 256, 0, 1 * 256>(c, fr, acc, gp, nullptr, relu_to(X), nothing);
  bx_segment<S, NH, BXS_L1, 16, 8, 0, 1 * 256, 0, 2 * 256>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 15));
  bx_segment<S, NH, BXS_L1 + 128, 16, 8, 0, 2 * 256, 0, 3 * 256>(c, fr, acc, Y, nullptr, relu_to(X), last_of(relu_to(Y), 15));
  bx_segment<S, NH, BXS_L1 + 256, 16, 8, 0, 3 * 256, 0, 4 * 256>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 15));
  bx_segment<S, NH, BXS_L4, 16, 8, 2, 4 * 256, 0, 5 * 256>(c, fr, acc, Y, gp, relu_to(X), last_of(relu_to(Y), 15));
  bx_segment<S, NH, BXS_L5, 16, 8, 0, 5 * 256, 0, 6 * 256>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 15));
  bx_segment<S, NH, BXS_L5 + 128, 16, 8, 0, 6 * 256, 0, 7 * 256>(c, fr, acc, Y, nullptr, relu_to(X), last_of(relu_to(Y), 15));
  bx_segment<S, NH, BXS_L5 + 256, 16, 8, 0, 7 * 256, 0, BXB_SIGMA>(c, fr, acc, X, nullptr, relu_to(Y), last_of(relu_to(X), 15));
  // ---- sigma head (one tile, row 0) on h7  (nerf.py:94, 113-115)
  float spre[NH] = {};
  auto sig_epi = [&](const Acc& A) {
#pragma unroll
    for (int h = 0; h < NH; ++h) spre[h] = A.c[h][0];
  };
  bx_segment<S, NH, BXS_SIG, 1, 8, 0, BXB_SIGMA, 0, BXB_DIR>(c, fr, acc, Y, nullptr, nothing_f, last_of(relu_to(Y), 15));
  // ---- point_info folded into dir_info: c = relu(W_dir[:, :24] gamma_d + W_fold h7 + bias) (nerf.py:117-118); its first tile also
  // retires the sigma tile
  {
    const int lane_d = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));  // (re-derived: no address register kept across the stream)
    const u32x4* const back = reinterpret_cast<const u32x4*>(lds + BF_LDS_BYTES) + NH * (c.wv * 64 + lane_d);
#pragma unroll
    for (int h = 0; h < NH; ++h) gd[h][0] = back[h];
  }
  bx_segment<S, NH, BXS_DIR, 8, 1, 8, BXB_DIR, 1, BXB_COL>(c, fr, acc, gd, Y, relu_to(X), sig_epi);
  mid(spre);
  // ---- colour head: rows 0..2 of one tile, sigmoid (nerf.py:99, 119)
  bx_segment<S, NH, BXS_COL, 1, 4, 0, BXB_COL, 1, -1>(c, fr, acc, X, nullptr, nothing_f, last_of(relu_to(X), 7));
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) cpre[h][ch] = acc[1].c[h][ch];
}

template <int NH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_field_fwd_bf16x(const FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using S = typename BxStreamOf<WAVES>::type;
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = c.lane, n = lane & 15;
  const int m0 = blockIdx.x * (16 * NH * WAVES) + c.wv * (16 * NH);

  // ---- ordinary loads first: this lane's NH samples (n, 16 + n, ... of the wave's 16 NH)
  int ms[NH];
  bool valid[NH];
  float p[NH][3], dw[NH][3];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    ms[h] = m0 + 16 * h + n;
    valid[h] = ms[h] < a.M;
    const int mc = valid[h] ? ms[h] : a.M - 1;
    const float* rf = a.rayf + (size_t)(mc / a.N) * RAYF;
    sample_point(rf, a.t[mc], p[h]);
#pragma unroll
    for (int i = 0; i < 3; ++i) dw[h][i] = rf[RF_DWRD + i];
  }
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(p[h][i]), "+v"(dw[h][i]));

  float cpre[NH][3];
  bx_field_pass<S, NH>(c, lds, p, dw, [&](const float (&spre)[NH]) {
    // (sample indices are re-derived from the lane id behind the stream -- mbcnt, not threadIdx: nothing to keep alive or spill)
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int n_e = lane_e & 15;
    const bool q0_e = lane_e < 16;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int me = m0 + 16 * h + n_e;
      if (me < a.M && q0_e) a.sigma[me] = fabsf(spre[h]);
    }
  }, cpre);
  const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int n_e = lane_e & 15;
  const bool q0_e = lane_e < 16;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int me = m0 + 16 * h + n_e;
    if (me < a.M && q0_e) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) a.rgb[(size_t)me * 3 + ch] = 1.0f / (1.0f + expf(-cpre[h][ch]));
    }
  }
}

// ------------------------------------------------------------------------------------------
// SMALL batches (a rank's share of a strong-scaling step, the reference's 400-ray batch): the whole inference forward of TWO rays in one
// workgroup and ONE launch -- coarse field pass (2 x 64 samples: 8 waves x 16), coarse composite + inverse-CDF resampling (two waves,
// one ray each, ray_parts.h), fine field pass (2 x 128 samples: 8 waves x 32), merge + five channel sorts + composite (two waves).
// As separate launches (field, k_coarse, field, k_merge) a 512-ray forward spends a third of its 0.12 ms in launch boundaries, cold
// starts and a coarse pass that fills half of the chip's waves; here every CU renders a ray pair end to end, the weight image streams
// through its LDS ring twice, and nothing but the ray records comes from -- and nothing but the two colours has to go to -- HBM (the
// per-sample outputs are still written: they are the workspace's introspection buffers, 7.7 KB per pair).
// Same arithmetic as the separate kernels: bx_field_pass is k_field_fwd_bf16x's body, the ray stages are ray_parts.h's functions --
// the pixels are bit-identical (tests/test_gpu_bf16.py::test_pair_kernel_equals_separate_launches).  Needs Nc = 64, Nf = 128.
// LDS: bias block [0, 16 KiB) whose tail behind the 70 bias tiles holds the pair's per-sample results, ring, parked encodings.
// ------------------------------------------------------------------------------------------
struct BxStreamNoBias : BxStream {
  static constexpr bool HAS_BIAS = false;  // second pass: the bias block is in LDS already (and its tail holds the first pass's results)
};
constexpr int PAIR_NC = 64, PAIR_NF = 128;
constexpr int PR_SIGC = 72 * 32;             // float offsets inside the bias block (tiles 0..69 are biases)
constexpr int PR_RGBC = PR_SIGC + 2 * PAIR_NC;
constexpr int PR_TF = PR_RGBC + 6 * PAIR_NC;
constexpr int PR_SIGF = PR_TF + 2 * PAIR_NF;
constexpr int PR_RGBF = PR_SIGF + 2 * PAIR_NF;
static_assert((PR_RGBF + 6 * PAIR_NF) * 4 <= BF_BIAS_BYTES && PR_SIGC >= 32 * BF_NBIAS_TILES, "results live behind the bias tiles");

// element `k` (lane-varying) of a register array without a scratch round trip: a select chain
__device__ __forceinline__ float rf_lane(const float (&rf)[RAYF], int k) {
  float v = rf[0];
#pragma unroll
  for (int j = 1; j < RAYF; ++j) v = (k == j) ? rf[j] : v;
  return v;
}
// flags |= bits in a word that carries the generation of the call it belongs to: (gen << 8) | flags.  A word of another generation (an
// earlier call's, or garbage) is replaced.  Rare (a ray met the reference's exit(0) condition), so the compare-and-swap loop costs nothing.
__device__ __forceinline__ void status_stamp_or(unsigned* word, unsigned gen, unsigned bits) {
  unsigned cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    const unsigned want = ((cur >> 8) == gen) ? (cur | bits) : ((gen << 8) | bits);
    if (__hip_atomic_compare_exchange_strong(word, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
}
__device__ __forceinline__ void wave_lds_fence() {  // this wave's LDS writes are visible to its own later reads (no workgroup barrier)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The lane id through the mbcnt BUILTINS (not bf16_stream.h's volatile asm): behind an MFMA the compiler may hand the asm a destination
// register that overlaps the accumulator of an MFMA still in flight (a dead row of it) -- the hazard recognizer does not look inside the
// asm, the MFMA's late write then clobbers the lane id (seen: every lane stored to sample 0).  With the builtins the wait states are
// inserted.
__device__ __forceinline__ int lane_id_builtin() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// The two field passes of the pair kernel.  The second one gets its lane id, LDS base and offsets through opaque values (fresh_ctx): left
// to itself the compiler shares address arithmetic between the two unrolled streams, keeps it alive across the first one, and the second
// pass -- which fills the 256 registers of a wave on its own -- spills 230 of them.
__device__ __forceinline__ BfCtx fresh_ctx(const BfCtx& c, unsigned char*& lds) {
  unsigned zero = 0;
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  BfCtx d;
  d.wimg = c.wimg;
  d.lds = lds + zero;
  d.lds_base = c.lds_base + zero;
  d.lane = (int)lane_id_here();
  d.wv = c.wv;
  lds = d.lds;
  return d;
}
__device__ __forceinline__ void pair_coarse_pass(const PairArgs& a, const BfCtx& c, unsigned char* lds, const int r0) {
  float* const res = reinterpret_cast<float*>(lds);
  // ================= coarse pass: sample s = 16 wv + n of the pair's 128 (ray s >> 6, depth index s & 63)
  {
    const int n = c.lane & 15;
    const int s = 16 * c.wv + n, rl = s >> 6, i = s & 63;
    const int ray = min(r0 + rl, a.B - 1);
    float rf[RAYF];  // the ray record in registers: what k_rays would have left in the workspace (same function, same bits)
    ray_record(a.rays, ray, rf);
    float p[1][3], dw[1][3];
    sample_point(rf, coarse_depth(rf[RF_NEAR], rf[RF_FAR], rf[RF_STEP], i, PAIR_NC), p[0]);
#pragma unroll
    for (int k = 0; k < 3; ++k) dw[0][k] = rf[RF_DWRD + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(p[0][k]), "+v"(dw[0][k]));
    float cpre[1][3];
    bx_field_pass<BxStream, 1>(c, lds, p, dw, [&](const float (&spre)[1]) {
      const int le = lane_id_builtin();
      if (le < 16) res[PR_SIGC + 16 * c.wv + le] = fabsf(spre[0]);
    }, cpre);
    const int le = lane_id_builtin();
    if (le < 16) {
      const int se = 16 * c.wv + le;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) res[PR_RGBC + 3 * se + ch] = 1.0f / (1.0f + expf(-cpre[0][ch]));
    }
  }
}

__device__ __forceinline__ void pair_fine_pass(const PairArgs& a, const BfCtx& c, unsigned char* lds, const int r0) {
  float* const res = reinterpret_cast<float*>(lds);
  // ================= fine pass: sample s = 32 wv + 16 h + n of the pair's 256 (ray s >> 7, depth index s & 127)
  {
    const int n = c.lane & 15;
    float p[2][3], dw[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = 32 * c.wv + 16 * h + n, rl = s >> 7;
      const int ray = min(r0 + rl, a.B - 1);
      float rf[RAYF];
      ray_record(a.rays, ray, rf);
      sample_point(rf, res[PR_TF + s], p[h]);
#pragma unroll
      for (int k = 0; k < 3; ++k) dw[h][k] = rf[RF_DWRD + k];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(p[h][k]), "+v"(dw[h][k]));
    float cpre[2][3];
    bx_field_pass<BxStreamNoBias, 2>(c, lds, p, dw, [&](const float (&spre)[2]) {
      const int le = lane_id_builtin();
      if (le < 16) {
#pragma unroll
        for (int h = 0; h < 2; ++h) res[PR_SIGF + 32 * c.wv + 16 * h + le] = fabsf(spre[h]);
      }
    }, cpre);
    const int le = lane_id_builtin();
    if (le < 16) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int se = 32 * c.wv + 16 * h + le;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) res[PR_RGBF + 3 * se + ch] = 1.0f / (1.0f + expf(-cpre[h][ch]));
      }
    }
  }
}

__global__ __launch_bounds__(BF_WG, 1) void k_render_pair_bf16x(const PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const res = reinterpret_cast<float*>(lds);
  const int r0 = 2 * blockIdx.x;  // this workgroup's rays r0, r0 + 1 (the second one may lie behind the batch: computed on a copy, not stored)
  // No kernel runs in front of this one in a rendering loop (the ray records are made here, the weight image is reused), so nobody has
  // zeroed the workspace's status word: this call's flags are STAMPED with its generation instead (common.h STATUS_*: the reader takes
  // them only if the stamp is this call's)
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.status) {
    a.status[STATUS_GEN_WORD] = a.gen;
    a.status[STATUS_SCHEME_WORD] = 1u;
  }

  pair_coarse_pass(a, c, lds, r0);
  __syncthreads();  // the pair's coarse sigma / rgb are in LDS; nobody reads the ring any more
  // ================= coarse composite + resampling: waves 0, 1 take one ray each; the ring's first bytes are their scratch
  if (c.wv < 2) {
    const int lane = lane_id_builtin();
    const int rl = c.wv, ray_raw = r0 + rl;
    const bool live = ray_raw < a.B;
    const int ray = live ? ray_raw : a.B - 1;
    float rf[RAYF];
    ray_record(a.rays, ray, rf);
    const float near = rf[RF_NEAR], far = rf[RF_FAR];
    // quirk Q6: the coarse spacing of the batch's ray 0 (pose row 0 of this call, or the caller's global ray 0)
    const float n0 = a.ray0_override ? a.near0 : a.rays.pb[15], f0 = a.ray0_override ? a.far0 : a.rays.pb[16];
    const float delta0 = ray0_spacing(n0, f0, PAIR_NC);
    float* scr = reinterpret_cast<float*>(lds + BF_BIAS_BYTES) + rl * 4 * PAIR_NC;
    float* sw = scr, *scdf = scr + PAIR_NC, *stc = scr + 2 * PAIR_NC, *stin = scr + 3 * PAIR_NC;
    const size_t g0 = (size_t)ray * PAIR_NC;
    stin[lane] = coarse_depth(near, far, rf[RF_STEP], lane, PAIR_NC);  // the ray's coarse depths (k_rays' t_c)
    wave_lds_fence();
    float lo, hi;
    coarse_ray_weights(res + PR_SIGC + PAIR_NC * rl, res + PR_RGBC + 3 * PAIR_NC * rl, stin, near, far, PAIR_NC, lane, sw, scdf, stc,
                       (live && a.w_c) ? a.w_c + g0 : nullptr, live ? a.C_coarse + (size_t)ray * 3 : nullptr, lo, hi);
    wave_lds_fence();
    const bool bad = coarse_ray_resample(sw, scdf, stc, lo, hi, delta0, PAIR_NC, PAIR_NF, lane, res + PR_TF + PAIR_NF * rl);
    if (live && bad && a.status) status_stamp_or(a.status + STATUS_STAMPED_WORD, a.gen, 1u);
    if (live && bad && a.sticky) atomicOr(a.sticky, 1u);
    wave_lds_fence();
    if (live) {  // the workspace's per-ray / per-sample buffers of the coarse pass and the fine depths (introspection; 2.9 KB per ray)
      if (lane < RAYF && a.rays.rayf) a.rays.rayf[(size_t)ray * RAYF + lane] = rf_lane(rf, lane);
      if (a.rays.t_c) a.rays.t_c[g0 + lane] = stin[lane];
      if (a.sig_c) a.sig_c[g0 + lane] = res[PR_SIGC + PAIR_NC * rl + lane];
      if (a.rgb_c)
#pragma unroll
        for (int k = 0; k < 3; ++k) a.rgb_c[g0 * 3 + 64 * k + lane] = res[PR_RGBC + 3 * PAIR_NC * rl + 64 * k + lane];
      if (a.t_f)
#pragma unroll
        for (int k = 0; k < 2; ++k) a.t_f[(size_t)ray * PAIR_NF + 64 * k + lane] = res[PR_TF + PAIR_NF * rl + 64 * k + lane];
    }
  }
  __syncthreads();  // the fine depths are in LDS; the scratch in the ring is free
  {
    unsigned char* lds2 = lds;
    const BfCtx c2 = fresh_ctx(c, lds2);
    pair_fine_pass(a, c2, lds2, r0);
  }
  __syncthreads();  // the pair's fine sigma / rgb are in LDS; the ring is free
  // ================= merge + five channel sorts + composite (nerf.py:302-321).  The pair's TEN channel sorts (2 rays x t, r, g, b, sigma: five
  // independent sorts per ray, quirk Q1) are dealt out over the eight waves -- each sorts one 256-slot channel in registers, the same network
  // as inside k_merge's five-channel sort -- and leaves it in val [2][5][256] in the ring; then waves 0, 1 composite one ray each.
  {
    constexpr int P = 256, N = PAIR_NC + PAIR_NF;
    float* const val_all = reinterpret_cast<float*>(lds + BF_BIAS_BYTES);
    const int lane = lane_id_builtin();
    for (int j = c.wv; j < 10; j += 8) {
      const int rl = j / 5, ch = j - 5 * rl;
      const int ray = min(r0 + rl, a.B - 1);
      const float nearm = a.rays.pb[(size_t)ray * 17 + 15], farm = a.rays.pb[(size_t)ray * 17 + 16];
      const float stepm = (farm - nearm) / (float)(PAIR_NC - 1);  // (= the ray record's RF_STEP)
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // slot i of the merged bundle: coarse samples first, then fine (k_merge's load), padding = NaN = the maximum key
        const int i = 4 * lane + r;
        float x;
        if (i < PAIR_NC) {
          const int sc = PAIR_NC * rl + i;
          x = ch == 0 ? coarse_depth(nearm, farm, stepm, i, PAIR_NC) : ch == 4 ? res[PR_SIGC + sc] : res[PR_RGBC + 3 * sc + (ch - 1)];
        } else if (i < N) {
          const int sf = PAIR_NF * rl + (i - PAIR_NC);
          x = ch == 0 ? res[PR_TF + sf] : ch == 4 ? res[PR_SIGF + sf] : res[PR_RGBF + 3 * sf + (ch - 1)];
        } else {
          x = __builtin_nanf("");
        }
        v[r] = x;
      }
      sort256_one_channel(v, lane);
      *reinterpret_cast<float4*>(val_all + (rl * 5 + ch) * P + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if (c.wv < 2) {
      const int rl = c.wv, ray_raw = r0 + rl;
      const bool live = ray_raw < a.B;
      const int ray = live ? ray_raw : a.B - 1;
      if (live) {  // the workspace's per-sample buffers of the fine pass (introspection)
        const size_t gf = (size_t)ray * PAIR_NF;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (a.sig_f) a.sig_f[gf + 64 * k + lane] = res[PR_SIGF + PAIR_NF * rl + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (a.rgb_f) a.rgb_f[gf * 3 + 64 * k + lane] = res[PR_RGBF + 3 * PAIR_NF * rl + 64 * k + lane];
      }
      float cf[3];
      float* const cout = live ? a.C_fine + (size_t)ray * 3 : cf;  // (a dead second ray: composited into registers nobody reads)
      merge_ray_composite<false>(val_all + rl * 5 * P, nullptr, P, N, a.last, lane, nullptr, nullptr, nullptr, cout);
    }
  }
}

hipError_t launch_render_pair_bf16x(const PairArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> opted{0};
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_render_pair_bf16x)}, BX_LDS_BYTES)) return e;
  hipLaunchKernelGGL(k_render_pair_bf16x, dim3((a.B + 1) / 2), dim3(BF_WG), BX_LDS_BYTES, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight image: fragment (tile T, k-step s), lane (i, q), slot j = W[16T + i][32s + 16 (j >> 2) + 4q + (j & 3)]
// ------------------------------------------------------------------------------------------
// the bias block is the 32x32x16 image's (same float layout); only the fragments differ
__global__ __launch_bounds__(256) void k_pack_weights_bf16x(const Weights24 w, const float* __restrict__ fold, unsigned char* __restrict__ img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= BX_NCHUNK * BF_CHUNK * 64) return;
  const int frag = gid >> 6, lane = gid & 63, i = lane & 15, q = lane >> 4;
  u32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // slots 2e, 2e + 1
    const int kk = 16 * (e >> 1) + 4 * q + 2 * (e & 1);
    v[e] = pack2(bx_weight(w, fold, frag, i, kk), bx_weight(w, fold, frag, i, kk + 1));
  }
  *reinterpret_cast<u32x4*>(img + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
}

hipError_t launch_pack_weights_bf16x(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  hipError_t e = launch_pack_bias_block_bf16(w, fold, img, st);
  if (e != hipSuccess) return e;
  const int threads = BX_NCHUNK * BF_CHUNK * 64;
  hipLaunchKernelGGL(k_pack_weights_bf16x, dim3((threads + 255) / 256), dim3(256), 0, st, w, fold, img);
  return hipGetLastError();
}

#ifndef NERF_BX_GROUPS  // 16-sample groups per wave: 2 = two waves per SIMD, 4 = the two-column form (make variant DEFS=-DNERF_BX_GROUPS=..)
#define NERF_BX_GROUPS 2
#endif
template <int NH, int WAVES>
static hipError_t bx_launch(const FieldArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> opted{0};
  constexpr int lds_bytes = BF_LDS_BYTES + 64 * WAVES * NH * 16, per_wg = 16 * NH * WAVES;
  static_assert(lds_bytes <= BX_LDS_BYTES, "LDS of one CU");
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_fwd_bf16x<NH, WAVES>)}, lds_bytes)) return e;
  hipLaunchKernelGGL((k_field_fwd_bf16x<NH, WAVES>), dim3((a.M + per_wg - 1) / per_wg), dim3(64 * WAVES), lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_field_fwd_bf16x(const FieldArgs& a, hipStream_t st) {
  constexpr int NH = NERF_BX_GROUPS;
  if constexpr (NH == 4) return bx_launch<4, 4>(a, st);
#ifndef NERF_BX_NO_SMALL
  // a pass of at most 256 x 128 samples: 4-wave workgroups, so that every CU gets one
  if ((a.M + 127) / 128 <= BX_SMALL_MAX_WGS) return bx_launch<2, 4>(a, st);
#endif
  return bx_launch<2, 8>(a, st);
}

}  // namespace nerf
