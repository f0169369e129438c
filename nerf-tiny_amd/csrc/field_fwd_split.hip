// field_fwd_split.hip -- INFERENCE forward of the field query with fp32 operands SPLIT into two bf16 parts (MI355X / gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate; v_mfma_f32_32x32x16_bf16 is 16x faster and its products of two bf16 values
// are exact in its fp32 accumulator.  Every fp32 operand x is written as x = hi + mid + (rest), hi = bf16(x), mid = bf16(x - hi):
// 16 significant bits, |rest| <= 2^-17 |x|.  A fp32 product then is
//        a b  =  a_hi b_hi + a_hi b_mid + a_mid b_hi  +  O(2^-16 |a b|)
// i.e. THREE bf16 MFMAs per k-step of 16 instead of eight fp32 MFMAs of k = 2: 96 instead of 512 MFMA cycles, fp32 accumulation,
// fp32 biases, fp32 everything else.  Measured against the reference's own outputs this evaluation sits where the exact-fp32 kernels
// sit (C_coarse 3e-6, C_fine 2e-5 max-rel on the golden cfg2 case: tests/test_gpu_split.py; the emulation of exactly this
// arithmetic in the build container: 3.3e-6 / 2.2e-5) -- the 1e-4 bar is met with the same margin, because what limits both is the
// conditioning of the fine pass, not the sixteenth bit of a product.  It is an OPT-IN mode (NERF_HIP_SPLIT_MLP, model.split_mlp /
// model.split_train): the default path keeps the exact k-ordered fp32 fma chains.
// SAVE (NERF_HIP_SPLIT_MLP | NERF_HIP_SAVE_FOR_BACKWARD: the split-fp32 TRAIN step, field_bwd_split.hip + dw_bf16.hip's SPLIT form): every layer input
// is also written to HBM as TWO bf16 fragment-layout tensors -- its hi parts and its mid parts, each in the layout of the bf16-MLP variant's
// save buffer (bf16_common.h: 1-KiB pieces, one per wave and k-step, whole lines per store) -- with the ReLU masks of the bf16 layout and the
// sigma pre-activation; a layer's 2 x 16 pieces + its mask words go out in one burst behind the layer's last tile.
//
// Machinery: bf16_stream.h (LDS ring of 1-KiB A fragments filled by direct-to-LDS loads, activations in registers, the accumulator
// of one layer = the operand of the next), in a 4-wave form: ONE wave per SIMD (the two-part activations of a layer's input and
// output are 2 x 128 registers), 128 samples per workgroup.  The stream interleaves the two parts of every fragment of the bf16
// layout of bf16_common.h (point_info folded into dir_info: common.h SEG_FOLD): fragment 2 s = hi, 2 s + 1 = mid of step s.
#include "bf16_stream.h"
#include "bf16_weights.h"

namespace nerf {

constexpr int SP_NFRAG = 2 * BF_NFRAG;           // 2,112 fragments = 132 chunks
constexpr int SP_NCHUNK = SP_NFRAG / BF_CHUNK;
static_assert(SP_NCHUNK * BF_CHUNK == SP_NFRAG, "whole chunks");
constexpr int SP_WG = 256;                        // 4 waves x 32 samples

// ---- store schedule of the SAVE variant (bf16_stream.h: stores share the vmcnt queue with the ring's loads and retire in order, so every
// counted wait allows for the younger ones).  Counted: the burst behind a layer's last tile -- 2 x (2 ntiles) pieces + 1 mask piece -- which
// every lane issues unconditionally.  The last tile's epilogue runs inside the NEXT segment's tile 0: spread over k-steps BF_EPI_POS ..
// BF_EPI_POS + 7 where that segment has >= BF_EPI_POS + 8 k-steps (the burst follows part 7), in one lump at k-step BF_EPI_POS otherwise.
// The conditional stores (sigma, spre, rgb) are NOT counted: an uncounted store only makes a wait more conservative.
struct SplitBurst { int next_s0, next_ks, stores; };  // next segment's first STEP (= bf16 fragment index) and k-steps per tile; stores of the burst
constexpr SplitBurst kSplitBursts[] = {
    {BFS_L1, 16, 33},       {BFS_L1 + 128, 16, 33}, {BFS_L1 + 256, 16, 33}, {BFS_L4, 20, 33}, {BFS_L5, 16, 33},
    {BFS_L5 + 128, 16, 33}, {BFS_L5 + 256, 16, 33}, {BFS_SIG, 16, 33},      {BFS_COL, 8, 17}};
struct SplitStoreTable { int cum[2 * BF_NFRAG + 1]; };
constexpr SplitStoreTable make_split_store_table() {
  SplitStoreTable t{};
  int ev[BF_NFRAG + 64] = {};  // counted stores issued in STEP i (after that step's sync point)
  for (const SplitBurst& b : kSplitBursts) ev[b.next_s0 + BF_EPI_POS + (b.next_ks >= BF_EPI_POS + 8 ? 7 : 0)] += b.stores;
  int run = 0, step = 0;
  for (int i = 0; i <= 2 * BF_NFRAG; ++i) {  // fragment index i: the steps before it are 0 .. i / 2 - 1 (sync points sit on even indices)
    while (step < i / 2) run += ev[step++];
    t.cum[i] = run;
  }
  return t;
}
constexpr SplitStoreTable kSplitStoreTable = make_split_store_table();

template <bool SAVE>
struct SplitStreamT {
  static constexpr int NFRAG = SP_NFRAG, NCHUNK = SP_NCHUNK, NS = BF_NS, RING_OFF = BF_BIAS_BYTES, D = 6, PW = 4;
  static constexpr bool HAS_BIAS = true;
  static constexpr int PROLOGUE_STORES = SAVE ? 12 : 0;  // gamma_p (4 k-steps) and gamma_d (2), hi and mid pieces each
  __device__ static constexpr int stores_before(int idx) { return SAVE ? kSplitStoreTable.cum[idx] : 0; }
};
using SplitStream = SplitStreamT<false>;
static_assert(BF_SYNC_POS % 2 == 0 && SplitStream::D % 2 == 0, "a step's two fragments stay in one chunk / keep their ring parity");

// two fp32 values -> their hi parts (packed bf16 pair) and mid parts
struct HiMid { unsigned hi, mid; };
__device__ __forceinline__ HiMid split2(float x0, float x1) {
  HiMid r;
  r.hi = pack2(x0, x1);
  const float h0 = __uint_as_float(r.hi << 16), h1 = __uint_as_float(r.hi & 0xffff0000u);
  r.mid = pack2(x0 - h0, x1 - h1);
  return r;
}

// One segment: NFT output tiles x (KSA + KSB) k-steps starting at STEP S0 (fragments 2 S0 ..); inputs inA (k-steps 0..KSA-1) then
// inB, each as hi / mid parts.  Per step three MFMAs, the small products first.  Accumulator hand-over as in bf_segment, except that
// the epilogue of the finished tile is SPREAD: with one wave per SIMD nothing else fills the matrix pipe while this wave converts, so the
// 8 register pairs of a tile (epi(f, part, acc): ~13 vector instructions each) go one per k-step behind k-steps 2 .. 9 of the next
// tile -- about four vector instructions per MFMA, which the 32-cycle MFMA hides -- instead of ~110 in one lump.  Segments shorter than
// ten k-steps keep the lump.
struct EpiTmp { float x0, x1; unsigned hi; };  // state of one register pair between the three phases of its conversion
// a ReLU'd value is >= +0: alive (its ReLU passes the gradient) <=> its bit pattern is >= 1; bit 15 - r of a tile's mask word = register r
__device__ __forceinline__ unsigned alive_pair(float x0, float x1, int r) {
  return ((unsigned)min(__float_as_int(x0), 1) << (15 - r)) | ((unsigned)min(__float_as_int(x1), 1) << (14 - r));
}

template <class S, int S0, int NFT, int KSA, int KSB, int BT0, int P0, int NEXT_BT, class Epi, class PrevEpi>
__device__ __forceinline__ void sp_segment(const BfCtx& c, u32x4 (&fr)[S::D], f32x16 (&acc)[2], const u32x4* inA_hi, const u32x4* inA_mid,
                                           const u32x4* inB_hi, const u32x4* inB_mid, Epi&& epi, PrevEpi&& prev_epi) {
  constexpr int KS = KSA + KSB;
  constexpr bool SPREAD = KS >= BF_EPI_POS + 8;
  constexpr int LAST = SPREAD ? BF_EPI_POS + 7 : BF_EPI_POS;  // k-step behind which the finished accumulator is free again
  static_assert(KS > LAST, "segment too short for the deferred epilogue");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  EpiTmp tmp = {0.f, 0.f, 0u};
  static_for<NFT * KS>([&](auto I) {
    constexpr int f = I / KS, ks = I % KS, i0 = 2 * (S0 + I), i1 = i0 + 1;
    constexpr int cur = (P0 + f) & 1, oth = (P0 + f + 1) & 1;
    constexpr bool EPI = SPREAD && ks >= BF_EPI_POS && ks <= LAST;
    auto phase = [&](auto PH) {  // one third of the conversion of register pair ks - BF_EPI_POS of the finished tile, behind ONE MFMA
      if constexpr (EPI) {
        if constexpr (f == 0)
          prev_epi(ks - BF_EPI_POS, (int)PH, acc[oth], tmp);
        else
          epi(f - 1, ks - BF_EPI_POS, (int)PH, acc[oth], tmp);
      }
    };
    if constexpr (i0 % BF_CHUNK == BF_SYNC_POS) bf_sync<S, i0 / BF_CHUNK>(c);
    const u32x4 a_hi = fr[i0 % S::D];
#ifndef NERF_TIMING_NO_FRAG  // (timing experiments only)
    if constexpr (i0 + S::D < S::NFRAG) fr[i0 % S::D] = bf_frag<S>(c, i0 + S::D);
#endif
    const u32x4 a_mid = fr[i1 % S::D];
#ifndef NERF_TIMING_NO_FRAG
    if constexpr (i1 + S::D < S::NFRAG) fr[i1 % S::D] = bf_frag<S>(c, i1 + S::D);
#endif
    const u32x4& b_hi = ks < KSA ? inA_hi[ks < KSA ? ks : 0] : inB_hi[ks < KSA ? 0 : ks - KSA];
    const u32x4& b_mid = ks < KSA ? inA_mid[ks < KSA ? ks : 0] : inB_mid[ks < KSA ? 0 : ks - KSA];
    // program order = issue order for one wave per SIMD: every MFMA is followed by its share of vector work, and the fences keep
    // the compiler from collecting that work into lumps the matrix pipe would have to wait for
    acc[cur] = bf_mfma(a_mid, b_hi, acc[cur]);
    phase(std::integral_constant<int, 0>{});
    if constexpr (EPI) __builtin_amdgcn_sched_barrier(0);
    acc[cur] = bf_mfma(a_hi, b_mid, acc[cur]);
    phase(std::integral_constant<int, 1>{});
    if constexpr (EPI) __builtin_amdgcn_sched_barrier(0);
    acc[cur] = bf_mfma(a_hi, b_hi, acc[cur]);
    phase(std::integral_constant<int, 2>{});
    if constexpr (EPI) __builtin_amdgcn_sched_barrier(0);
    if constexpr (!SPREAD && ks == BF_EPI_POS) {
      static_for<8>([&](auto P) {
        static_for<3>([&](auto PH) {
          if constexpr (f == 0)
            prev_epi((int)P, (int)PH, acc[oth], tmp);
          else
            epi(f - 1, (int)P, (int)PH, acc[oth], tmp);
        });
      });
    }
    if constexpr (ks == LAST) {
      if constexpr (f + 1 < NFT)
        acc[oth] = BT0 >= 0 ? bf_bias_tile(c, BT0 + f + 1) : zero;
      else if constexpr (NEXT_BT >= 0)
        acc[oth] = bf_bias_tile(c, NEXT_BT);
      else
        acc[oth] = zero;
    }
  });
}

template <bool SAVE>
__global__ __launch_bounds__(SP_WG, 1) void k_field_fwd_split(const FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using S = SplitStreamT<SAVE>;
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = c.lane, j = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * (SP_WG / 2) + c.wv * 32 + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int ray = mc / a.N;

  // ---- ordinary loads first (drained before anything else is in flight)
  const float* rf = a.rayf + (size_t)ray * RAYF;
  float p[3], dw[3];
  sample_point(rf, a.t[mc], p);
#pragma unroll
  for (int i = 0; i < 3; ++i) dw[i] = rf[RF_DWRD + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(p[i]), "+v"(dw[i]));

  bf_stream_start<S>(c);

  // ---- positional encodings (fp32, as in the exact path) straight into two-part B operands:
  // k-step ks, slot pair (s, s+1): features k = 16ks + 4h + {0,1 | 2,3 | 8,9 | 10,11} = (sin, cos) pairs pi = 8ks + 2h + {0, 1, 4, 5}
  u32x4 gp_hi[4], gp_mid[4], gd_hi[2], gd_mid[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pi = 8 * ks + 2 * h + (q & 1) + 4 * (q >> 1);
      float sv = 0.f, cv = 0.f;
      if (pi < 30) {
        const int cc = pi / 10, l = pi - 10 * cc;
        const float x = (cc == 0) ? p[0] : ((cc == 1) ? p[1] : p[2]);
#ifdef NERF_TIMING_NO_SINCOS  // (timing experiments only)
        sv = x; cv = x * 0.5f;
#else
        sincos_phase(x * __uint_as_float(kFreqPointBits[l]), sv, cv);
#endif
      }
      const HiMid e = split2(sv, cv);
      gp_hi[ks][q] = e.hi;
      gp_mid[ks][q] = e.mid;
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
#ifdef NERF_TIMING_NO_SINCOS
        sv = x; cv = x * 0.5f;
#else
        sincos_phase(x * __uint_as_float(kFreqDirBits[l]), sv, cv);
#endif
      }
      const HiMid e = split2(sv, cv);
      gd_hi[ks][q] = e.hi;
      gd_mid[ks][q] = e.mid;
    }

  // ---- training: this wave's block of the two fragment-layout save buffers (hi parts: a.bsave, mid parts: a.bsave2; bf16_common.h);
  // every lane stores, also lanes beyond the pass (they hold a copy of the last sample): the counted waits rely on the stores being issued
  const int wb = a.wb0 + blockIdx.x * (SP_WG / 64) + c.wv;
  unsigned mw[4];  // mask words of the layer being finished, two tiles per register (every word is assigned by its even tile before the odd one ORs in)
  auto save_pair = [&](int tensor, int ks, const u32x4& vh, const u32x4& vm, unsigned lane16) {
    const size_t off = ((size_t)a.wb_tot * bs_cum(tensor) + (size_t)wb * bs_ks(tensor) + ks) * BF_FRAG_BYTES + lane16;
    store_piece(a.bsave + off, vh);
    store_piece(a.bsave2 + off, vm);
  };
  if (SAVE) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) save_pair(BS_GP, ks, gp_hi[ks], gp_mid[ks], lane * 16);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) save_pair(BS_GD, ks, gd_hi[ks], gd_mid[ks], lane * 16);
  }

  u32x4 fr[S::D];
  bf_stream_first<S>(c, fr);

  u32x4 Xh[16], Xm[16], Yh[16], Ym[16];
  f32x16 acc[2];
  acc[0] = bf_bias_tile(c, BFB_L0);
  // epilogue of a ReLU layer, one register pair group at a time: part = (mh, q) of tile f (fp32 accumulators) -> ReLU (one integer max
  // per value: the fp32 bit pattern of max(x, 0)) -> two-part packed slot q of k-step 2f + mh of the next layer's input
  // SAVE: tensor >= 0 -> behind the layer's LAST tile (ntiles - 1) its whole output goes to the two save buffers in one burst, with the layer's
  // mask piece (u16 [layer][wave block][lane][8 tiles], bit 15 - r = register r alive: the bf16 variant's layout)
  auto relu_to = [&](u32x4* oh, u32x4* om, int tensor = -1, int mlayer = -1, int ntiles = 8) {
    return [&, oh, om, tensor, mlayer, ntiles](int f, int part, int ph, const f32x16& A, EpiTmp& t) {
      const int mh = part >> 2, q = part & 3;
      if (ph == 0) {  // ReLU (integer max on the fp32 bit patterns), hi parts
        t.x0 = __int_as_float(max(__float_as_int(A[8 * mh + 2 * q]), 0));
        t.x1 = __int_as_float(max(__float_as_int(A[8 * mh + 2 * q + 1]), 0));
        t.hi = pack2(t.x0, t.x1);
        oh[2 * f + mh][q] = t.hi;
        if (SAVE && mlayer >= 0) {
          const unsigned bits = alive_pair(t.x0, t.x1, 8 * mh + 2 * q) << (16 * (f & 1));
          mw[f >> 1] = (part == 0 && (f & 1) == 0) ? bits : (mw[f >> 1] | bits);
        }
      } else if (ph == 1) {  // residuals
        t.x0 -= __uint_as_float(t.hi << 16);
        t.x1 -= __uint_as_float(t.hi & 0xffff0000u);
      } else {  // mid parts
        om[2 * f + mh][q] = pack2(t.x0, t.x1);
        if (SAVE && tensor >= 0 && f == ntiles - 1 && part == 7) {  // the layer's output is complete: one contiguous burst per buffer
          const unsigned lane16 = 16u * lane_id_here();
          if (mlayer >= 0) {
            const u32x4 mv = {mw[0], mw[1], ntiles > 4 ? mw[2] : 0u, ntiles > 4 ? mw[3] : 0u};
            store_piece(reinterpret_cast<unsigned char*>(a.bmask) + ((size_t)mlayer * a.wb_tot + wb) * 1024 + lane16, mv);
          }
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
            if (ks < 2 * ntiles) save_pair(tensor, ks, oh[ks], om[ks], lane16);
        }
      }
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](int part, int ph, const f32x16& A, EpiTmp& t) { epi(f, part, ph, A, t); }; };
  auto nothing = [](int, int, const f32x16&, EpiTmp&) {};
  auto nothing_f = [](int, int, int, const f32x16&, EpiTmp&) {};

  // ---- layers 0..7 (nerf.py:104-112); e_l turns layer l's accumulators into h_l (two-part operands of layer l + 1; SAVE: + its burst)
  auto e0 = relu_to(Xh, Xm, BS_H0 + 0, 0), e1 = relu_to(Yh, Ym, BS_H0 + 1, 1), e2 = relu_to(Xh, Xm, BS_H0 + 2, 2), e3 = relu_to(Yh, Ym, BS_H0 + 3, 3);
  auto e4 = relu_to(Xh, Xm, BS_H0 + 4, 4), e5 = relu_to(Yh, Ym, BS_H0 + 5, 5), e6 = relu_to(Xh, Xm, BS_H0 + 6, 6), e7 = relu_to(Yh, Ym, BS_H0 + 7, 7);
  auto ec = relu_to(Xh, Xm, BS_C, 8, 4);  // c = relu(dir_info pre-activation): 4 tiles
  sp_segment<S, BFS_L0, 8, 4, 0, BFB_L0, 0, BFB_L0 + 8>(c, fr, acc, gp_hi, gp_mid, nullptr, nullptr, e0, nothing);
  sp_segment<S, BFS_L1, 8, 16, 0, BFB_L0 + 8, 0, BFB_L0 + 16>(c, fr, acc, Xh, Xm, nullptr, nullptr, e1, last_of(e0, 7));
  sp_segment<S, BFS_L1 + 128, 8, 16, 0, BFB_L0 + 16, 0, BFB_L0 + 24>(c, fr, acc, Yh, Ym, nullptr, nullptr, e2, last_of(e1, 7));
  sp_segment<S, BFS_L1 + 256, 8, 16, 0, BFB_L0 + 24, 0, BFB_L0 + 32>(c, fr, acc, Xh, Xm, nullptr, nullptr, e3, last_of(e2, 7));
  sp_segment<S, BFS_L4, 8, 16, 4, BFB_L0 + 32, 0, BFB_L0 + 40>(c, fr, acc, Yh, Ym, gp_hi, gp_mid, e4, last_of(e3, 7));
  sp_segment<S, BFS_L5, 8, 16, 0, BFB_L0 + 40, 0, BFB_L0 + 48>(c, fr, acc, Xh, Xm, nullptr, nullptr, e5, last_of(e4, 7));
  sp_segment<S, BFS_L5 + 128, 8, 16, 0, BFB_L0 + 48, 0, BFB_L0 + 56>(c, fr, acc, Yh, Ym, nullptr, nullptr, e6, last_of(e5, 7));
  sp_segment<S, BFS_L5 + 256, 8, 16, 0, BFB_L0 + 56, 0, BFB_SIGMA>(c, fr, acc, Xh, Xm, nullptr, nullptr, e7, last_of(e6, 7));
  // ---- sigma head (one tile, row 0) on h7: sigma = |w_sigma . h7 + b|  (nerf.py:94, 113-115)
  float spre = 0.f;
  auto sig_epi = [&](int part, int ph, const f32x16& A, EpiTmp&) { if (part == 0 && ph == 0) spre = A[0]; };
  sp_segment<S, BFS_SIG, 1, 16, 0, BFB_SIGMA, 0, BFB_DIR>(c, fr, acc, Yh, Ym, nullptr, nullptr, nothing_f, last_of(e7, 7));
  // ---- point_info folded into dir_info: c = relu(W_dir[:, :24] gamma_d + W_fold h7 + b_dir + W_dir[:, 24:] b_pi)  (nerf.py:117-118)
  sp_segment<S, BFS_DIR, 4, 2, 16, BFB_DIR, 1, BFB_COL>(c, fr, acc, gd_hi, gd_mid, Yh, Ym, ec, sig_epi);
  if (SAVE && valid && h == 0) a.spre[a.row0 + m] = spre;
  if (valid && h == 0) a.sigma[m] = fabsf(spre);
  // ---- colour head: rows 0..2 of one tile, sigmoid (nerf.py:99, 119)
  sp_segment<S, BFS_COL, 1, 8, 0, BFB_COL, 1, -1>(c, fr, acc, Xh, Xm, nullptr, nullptr, nothing_f, last_of(ec, 3));
  if (valid && h == 0) {
    a.rgb[(size_t)m * 3 + 0] = 1.0f / (1.0f + expf(-acc[1][0]));
    a.rgb[(size_t)m * 3 + 1] = 1.0f / (1.0f + expf(-acc[1][1]));
    a.rgb[(size_t)m * 3 + 2] = 1.0f / (1.0f + expf(-acc[1][2]));
  }
}

// ------------------------------------------------------------------------------------------
// weight image: bias block (fp32, the bf16 image's), then for every step s of the bf16 layout the hi fragment and the mid fragment
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights_split(const Weights24 w, const float* __restrict__ fold, unsigned char* __restrict__ img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid < BF_NFRAG * 64) {
    const int step = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
    u32x4 vh, vm;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
      const HiMid e = split2(bf_weight(w, fold, step, i, kk, h), bf_weight(w, fold, step, i, kk + 1, h));
      vh[q] = e.hi;
      vm[q] = e.mid;
    }
    unsigned char* dst = img + BF_BIAS_BYTES + (size_t)(2 * step) * BF_FRAG_BYTES + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = vh;
    *reinterpret_cast<u32x4*>(dst + BF_FRAG_BYTES) = vm;
  } else {
    const int b = gid - BF_NFRAG * 64;
    if (b < BF_BIAS_BYTES / 4) {
      const int tile = b >> 5, i = b & 31;
      reinterpret_cast<float*>(img)[b] = tile < BF_NBIAS_TILES ? bf_bias(w, fold, tile, i) : 0.f;
    }
  }
}

size_t split_image_bytes() { return (size_t)BF_BIAS_BYTES + (size_t)SP_NFRAG * BF_FRAG_BYTES; }

hipError_t launch_pack_weights_split(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  const int threads = BF_NFRAG * 64 + BF_BIAS_BYTES / 4;
  hipLaunchKernelGGL(k_pack_weights_split, dim3((threads + 255) / 256), dim3(256), 0, st, w, fold, img);
  return hipGetLastError();
}

hipError_t launch_field_fwd_split(const FieldArgs& a, bool save, hipStream_t st) {
  static std::atomic<unsigned long long> opted{0}, opted_save{0};
  int wgs = (a.M + SP_WG / 2 - 1) / (SP_WG / 2);
  // training: the fragment-layout buffers count wave blocks in whole 256-sample groups (api.hip wave_blocks); EVERY one of them is read by the
  // weight-gradient products, so every one is written -- an odd workgroup count gets a trailing workgroup of lanes past the end (copies of the
  // last sample forward, exact zeros in the chain's gradients)
  if (save) wgs = (wgs + 1) / 2 * 2;
  if (save) {
    if (!a.bsave || !a.bsave2 || !a.bmask || !a.spre) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds(opted_save, {reinterpret_cast<const void*>(&k_field_fwd_split<true>)}, BF_LDS_BYTES)) return e;
    hipLaunchKernelGGL(k_field_fwd_split<true>, dim3(wgs), dim3(SP_WG), BF_LDS_BYTES, st, a);
  } else {
    if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_fwd_split<false>)}, BF_LDS_BYTES)) return e;
    hipLaunchKernelGGL(k_field_fwd_split<false>, dim3(wgs), dim3(SP_WG), BF_LDS_BYTES, st, a);
  }
  return hipGetLastError();
}

}  // namespace nerf
