// field_bwd_split.hip -- backward dX chain of the split-fp32 TRAIN step (NERF_HIP_SPLIT_MLP | NERF_HIP_SAVE_FOR_BACKWARD), MI355X / gfx950.
// (forward: field_fwd_split.hip<SAVE>; weight gradients: the SPLIT instantiations of dw_bf16.hip)
//
// field_bwd_bf16.hip with every fp32 operand written as hi + mid (two bf16 parts, field_fwd_split.hip): the stream holds the TRANSPOSED
// weights from the colour head back to layer 0 as (hi, mid) fragment pairs, a wave owns 32 samples, and the fp32 accumulator of "d input" of
// one layer, masked with the layer's saved ReLU bits and split into hi + mid, is the two-part B operand of the next (earlier) layer:
//     dz -> dc -> dpre_dir -> dh7 (W_fold^T + w_sigma dspre) -> dpre7 -> ... -> dpre0 [-> d gamma_p -> dt, fine pass]
// three bf16 MFMAs per k-step (mid x hi, hi x mid, hi x hi), fp32 accumulation.  Every masked accumulator (= pre-activation gradient, the A
// operand of that layer's weight-gradient products) is written to TWO gradient buffers in fragment layout (bf16_common.h) -- hi parts to
// a.bG, mid parts to a.bG2 -- a layer's 2 x 16 pieces in one burst behind its last tile.  One wave per SIMD (the two-part operands of a
// layer's input and output are 2 x 128 registers), 128 samples per workgroup; the wave's ReLU masks (9 x 1 KiB) come to LDS once, ahead of
// the stream, by the same direct-to-LDS loads (LDS: 36 KiB masks + 7 x 16 KiB ring).
#include "bf16_stream.h"
#include "bf16_weights.h"

namespace nerf {

constexpr int SPB_WG = 256;                                  // 4 waves x 32 samples
constexpr int SPB_NS = 7;
constexpr int SPB_MASK_BYTES = 4 * BM_LAYERS * 1024;         // per workgroup: wave w, layer l at (w * 9 + l) * 1024
constexpr int SPB_LDS_BYTES = SPB_MASK_BYTES + SPB_NS * BF_CHUNK * BF_FRAG_BYTES;
static_assert(SPB_LDS_BYTES <= 160 * 1024, "LDS of one CU");
static_assert((2 * BBC_NFRAG) % BF_CHUNK == 0 && (2 * BBF_NFRAG) % BF_CHUNK == 0, "whole chunks");

// stores of the bursts (2 x (2 ntiles) pieces), each issued inside tile 0 of the NEXT segment: lumped at k-step BF_EPI_POS where that
// segment has < BF_EPI_POS + 8 k-steps, behind part 7 (k-step BF_EPI_POS + 7) otherwise.  STEPS = bf16 fragment indices.
struct SplitBwdBurst { int next_s0, next_ks, stores; };
template <bool FINE>
struct SplitBwdStream {
  static constexpr int NSTEP = FINE ? BBF_NFRAG : BBC_NFRAG;
  static constexpr int NFRAG = 2 * NSTEP, NCHUNK = NFRAG / BF_CHUNK;
  static constexpr int NS = SPB_NS, RING_OFF = SPB_MASK_BYTES, D = 6, PW = 4;
  static constexpr bool HAS_BIAS = false;
  static constexpr int PROLOGUE_STORES = 4;  // the dz / dspre fragment and its zero partner, hi and mid
  struct Table { int cum[2 * BBF_NFRAG + 1]; };
  static constexpr Table make() {
    const SplitBwdBurst bursts[] = {{BBS_FOLDT, 9, 16},      {BBS_L7T, 16, 32},       {BBS_L7T + 128, 16, 32}, {BBS_L7T + 256, 16, 32}, {BBS_L4T, 16, 32},
                                    {BBS_L3T, 16, 32},       {BBS_L3T + 128, 16, 32}, {BBS_L3T + 256, 16, 32}, {BBS_G0T, 32, 32}};  // (the last one: fine pass only)
    Table t{};
    int ev[BBF_NFRAG + 64] = {};
    for (const SplitBwdBurst& b : bursts) {
      const int e = b.next_s0 + BF_EPI_POS + (b.next_ks >= BF_EPI_POS + 8 ? 7 : 0);
      if (e < NSTEP) ev[e] += b.stores;
    }
    int run = 0, step = 0;
    for (int i = 0; i <= NFRAG; ++i) {
      while (step < i / 2) run += ev[step++];
      t.cum[i] = run;
    }
    return t;
  }
  static constexpr Table tab = make();
  __device__ static constexpr int stores_before(int idx) { return tab.cum[idx]; }
};

struct HiMidB { unsigned hi, mid; };
__device__ __forceinline__ HiMidB split2b(float x0, float x1) {
  HiMidB r;
  r.hi = pack2(x0, x1);
  const float h0 = __uint_as_float(r.hi << 16), h1 = __uint_as_float(r.hi & 0xffff0000u);
  r.mid = pack2(x0 - h0, x1 - h1);
  return r;
}

struct EpiTmpB { float x0, x1; unsigned hi; };

// One segment of the two-part stream (field_fwd_split.hip's sp_segment without biases): NFT output tiles x (KSA + KSB) k-steps starting at
// STEP S0; inputs inA then inB as hi / mid parts; the finished tile's epilogue spread one register pair per k-step where the tile is long enough.
template <class S, int S0, int NFT, int KSA, int KSB, int P0, class Epi, class PrevEpi>
__device__ __forceinline__ void spb_segment(const BfCtx& c, u32x4 (&fr)[S::D], f32x16 (&acc)[2], const u32x4* inA_hi, const u32x4* inA_mid,
                                            const u32x4* inB_hi, const u32x4* inB_mid, Epi&& epi, PrevEpi&& prev_epi) {
  constexpr int KS = KSA + KSB;
  constexpr bool SPREAD = KS >= BF_EPI_POS + 8;
  constexpr int LAST = SPREAD ? BF_EPI_POS + 7 : BF_EPI_POS;
  static_assert(KS > LAST, "segment too short for the deferred epilogue");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  EpiTmpB tmp = {0.f, 0.f, 0u};
  static_for<NFT * KS>([&](auto I) {
    constexpr int f = I / KS, ks = I % KS, i0 = 2 * (S0 + I), i1 = i0 + 1;
    constexpr int cur = (P0 + f) & 1, oth = (P0 + f + 1) & 1;
    constexpr bool EPI = SPREAD && ks >= BF_EPI_POS && ks <= LAST;
    auto phase = [&](auto PH) {
      if constexpr (EPI) {
        if constexpr (f == 0)
          prev_epi(ks - BF_EPI_POS, (int)PH, acc[oth], tmp);
        else
          epi(f - 1, ks - BF_EPI_POS, (int)PH, acc[oth], tmp);
      }
    };
    if constexpr (i0 % BF_CHUNK == BF_SYNC_POS) bf_sync<S, i0 / BF_CHUNK>(c);
    const u32x4 a_hi = fr[i0 % S::D];
    if constexpr (i0 + S::D < S::NFRAG) fr[i0 % S::D] = bf_frag<S>(c, i0 + S::D);
    const u32x4 a_mid = fr[i1 % S::D];
    if constexpr (i1 + S::D < S::NFRAG) fr[i1 % S::D] = bf_frag<S>(c, i1 + S::D);
    const u32x4& b_hi = ks < KSA ? inA_hi[ks < KSA ? ks : 0] : inB_hi[ks < KSA ? 0 : ks - KSA];
    const u32x4& b_mid = ks < KSA ? inA_mid[ks < KSA ? ks : 0] : inB_mid[ks < KSA ? 0 : ks - KSA];
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
    if constexpr (ks == LAST) acc[oth] = zero;
  });
}

template <bool FINE>
__global__ __launch_bounds__(SPB_WG, 1) void k_field_bwd_split(const FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using S = SplitBwdStream<FINE>;
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = c.lane, j = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * (SPB_WG / 2) + c.wv * 32 + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int wb = a.wb0 + blockIdx.x * (SPB_WG / 64) + c.wv;

  // ---- ordinary loads first: upstream gradients -> dz (colour head, pre-sigmoid) and dspre (sigma head, pre-abs)
  float dz[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float o = a.rgb[(size_t)mc * 3 + ch];
    dz[ch] = valid ? a.drgb[(size_t)mc * 3 + ch] * ((1.0f - o) * o) : 0.f;
  }
  const float sp = a.spre[a.row0 + mc];
  const float sgn = sp > 0.f ? 1.0f : (sp < 0.f ? -1.0f : 0.f);  // d|x|/dx with sign(0) = 0 like torch
  const float ds = valid ? a.dsig[mc] * sgn : 0.f;
  // as a two-part B operand / gradient fragment: inputs 0..2 = dz, input 3 = dspre  (k = 4h + s: lane half 0, slots 0..3)
  u32x4 zh[4], zm[4];
  {
    const HiMidB e0 = split2b(dz[0], dz[1]), e1 = split2b(dz[2], ds);
    zh[0] = u32x4{h == 0 ? e0.hi : 0u, h == 0 ? e1.hi : 0u, 0u, 0u};
    zm[0] = u32x4{h == 0 ? e0.mid : 0u, h == 0 ? e1.mid : 0u, 0u, 0u};
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(zh[0][q]), "+v"(zm[0][q]));  // the loads above are complete from here on
  zh[1] = zh[2] = zh[3] = zm[1] = zm[2] = zm[3] = u32x4{0u, 0u, 0u, 0u};

  // ---- this wave's ReLU masks -> LDS (9 x 1 KiB), then the weight stream
#pragma unroll
  for (int l = 0; l < BM_LAYERS; ++l)
    glds16(reinterpret_cast<const unsigned char*>(a.bmask) + ((size_t)l * a.wb_tot + wb) * 1024 + lane * 16,
           c.lds_base + (c.wv * BM_LAYERS + l) * 1024);
  bf_stream_start<S>(c);

  auto grad_pair = [&](int tensor, int ks, const u32x4& vh, const u32x4& vm, unsigned lane16) {
    const size_t off = ((size_t)a.wb_tot * bg_cum(tensor) + (size_t)wb * bg_ks(tensor) + ks) * BF_FRAG_BYTES + lane16;
    store_piece(a.bG + off, vh);
    store_piece(a.bG2 + off, vm);
  };
  grad_pair(BG_Z, 0, zh[0], zm[0], lane * 16);
  grad_pair(BG_Z, 1, zh[1], zm[1], lane * 16);

  u32x4 fr[S::D];
  bf_stream_first<S>(c, fr);

  u32x4 Xh[16], Xm[16], Yh[16], Ym[16];
  f32x16 acc[2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc[0] = zero;
  const uint16_t* const mk = reinterpret_cast<const uint16_t*>(lds + c.wv * BM_LAYERS * 1024) + lane * 8;  // [layer][lane][8 tiles]
  // epilogue, one register pair (part = (mh, q)) of d(input) tile f per call and phase: mask with the ReLU bits of `mlayer` (bit 15 - r = register r
  // alive) -> two-part packed slot q of k-step 2f + mh of the next GEMM; behind the tensor's last tile its 2 x (2 ntiles) pieces in one burst
  auto grad_to = [&](u32x4* oh, u32x4* om, int tensor, int mlayer, int ntiles = 8) {
    return [&, oh, om, tensor, mlayer, ntiles](int f, int part, int ph, const f32x16& A, EpiTmpB& t) {
      const int mh = part >> 2, q = part & 3, r = 8 * mh + 2 * q;
      if (ph == 0) {
        const int bits = mk[mlayer * 512 + f];
        t.x0 = __uint_as_float(__float_as_uint(A[r]) & (unsigned)__builtin_amdgcn_sbfe(bits, 15 - r, 1));
        t.x1 = __uint_as_float(__float_as_uint(A[r + 1]) & (unsigned)__builtin_amdgcn_sbfe(bits, 14 - r, 1));
        t.hi = pack2(t.x0, t.x1);
        oh[2 * f + mh][q] = t.hi;
      } else if (ph == 1) {
        t.x0 -= __uint_as_float(t.hi << 16);
        t.x1 -= __uint_as_float(t.hi & 0xffff0000u);
      } else {
        om[2 * f + mh][q] = pack2(t.x0, t.x1);
        if (f == ntiles - 1 && part == 7) {
          const unsigned lane16 = 16u * lane_id_here();
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
            if (ks < 2 * ntiles) grad_pair(tensor, ks, oh[ks], om[ks], lane16);
        }
      }
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](int part, int ph, const f32x16& A, EpiTmpB& t) { epi(f, part, ph, A, t); }; };
  auto nothing = [](int, int, const f32x16&, EpiTmpB&) {};

  auto gd = grad_to(Xh, Xm, BG_D, 8, 4);  // d c through c's ReLU = dpre_dir (4 tiles)
  auto g7 = grad_to(Yh, Ym, BG_L0 + 7, 7), g6 = grad_to(Xh, Xm, BG_L0 + 6, 6), g5 = grad_to(Yh, Ym, BG_L0 + 5, 5), g4 = grad_to(Xh, Xm, BG_L0 + 4, 4);
  auto g3 = grad_to(Yh, Ym, BG_L0 + 3, 3), g2 = grad_to(Xh, Xm, BG_L0 + 2, 2), g1 = grad_to(Yh, Ym, BG_L0 + 1, 1), g0 = grad_to(Xh, Xm, BG_L0 + 0, 0);
  // colour head: dc = W_color^T dz, through c's ReLU -> dpre_dir
  spb_segment<S, BBS_COLT, 4, 4, 0, 0>(c, fr, acc, zh, zm, nullptr, nullptr, gd, nothing);
  // dir_info and point_info as ONE transposed layer (folded) + the sigma head: dh7 = W_fold^T dpre_dir + w_sigma dspre, through h7's ReLU
  spb_segment<S, BBS_FOLDT, 8, 8, 1, 0>(c, fr, acc, Xh, Xm, zh, zm, g7, last_of(gd, 3));
  spb_segment<S, BBS_L7T, 8, 16, 0, 0>(c, fr, acc, Yh, Ym, nullptr, nullptr, g6, last_of(g7, 7));
  spb_segment<S, BBS_L7T + 128, 8, 16, 0, 0>(c, fr, acc, Xh, Xm, nullptr, nullptr, g5, last_of(g6, 7));
  spb_segment<S, BBS_L7T + 256, 8, 16, 0, 0>(c, fr, acc, Yh, Ym, nullptr, nullptr, g4, last_of(g5, 7));
  spb_segment<S, BBS_L4T, 8, 16, 0, 0>(c, fr, acc, Xh, Xm, nullptr, nullptr, g3, last_of(g4, 7));
  spb_segment<S, BBS_L3T, 8, 16, 0, 0>(c, fr, acc, Yh, Ym, nullptr, nullptr, g2, last_of(g3, 7));
  spb_segment<S, BBS_L3T + 128, 8, 16, 0, 0>(c, fr, acc, Xh, Xm, nullptr, nullptr, g1, last_of(g2, 7));
  spb_segment<S, BBS_L3T + 256, 8, 16, 0, 0>(c, fr, acc, Yh, Ym, nullptr, nullptr, g0, last_of(g1, 7));
  if constexpr (!FINE) {
    EpiTmpB t = {0.f, 0.f, 0u};
    static_for<8>([&](auto P) { static_for<3>([&](auto PH) { g0(7, (int)P, (int)PH, acc[1], t); }); });  // the last tile of the stream
  } else {
    // ---- d gamma_p (fp32) = W_0^T dpre0 + W_4[:, 256:]^T dpre4  (nerf.py:104, 109).  dpre4 is long gone from the registers: the wave reads
    // back the 2 x 16 pieces it stored itself (complete: every counted wait since has retired them; nobody on this CU has read those lines),
    // while G0T runs.  One segment: tile f of d gamma_p over the 16 k-steps of dpre0 (X) and then the 16 of dpre4 (Y) in one accumulator.
    f32x16 dgp[2];
    {
      const unsigned lane16 = 16u * lane_id_here();
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const size_t off = ((size_t)a.wb_tot * bg_cum(BG_L0 + 4) + (size_t)wb * 16 + ks) * BF_FRAG_BYTES + lane16;
        Yh[ks] = *reinterpret_cast<const u32x4*>(a.bG + off);
        Ym[ks] = *reinterpret_cast<const u32x4*>(a.bG2 + off);
      }
    }
    auto take0 = [&](int, int part, int ph, const f32x16& A, EpiTmpB&) { if (part == 0 && ph == 0) dgp[0] = A; };
    spb_segment<S, BBS_G0T, 2, 16, 16, 0>(c, fr, acc, Xh, Xm, Yh, Ym, take0, last_of(g0, 7));
    dgp[1] = acc[1];
    // gamma -> point -> depth (t_fine is not detached, quirk Q9).  dgp[t][4g + 2e], [.. + 1] = d loss / d (sin, cos) of pair pi = 16t + 4g + 2h + e
    const int lane_e = (int)lane_id_here();
    const int h_e = lane_e >> 5;
    const int m_e = blockIdx.x * (SPB_WG / 2) + c.wv * 32 + (lane_e & 31);
    const bool valid_e = m_e < a.M;
    const int mcl = valid_e ? m_e : a.M - 1;
    const int ray = mcl / a.N;
    const float* rf = a.rayf + (size_t)ray * RAYF;
    float p[3];
    sample_point(rf, a.t[mcl], p);
    float dp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pi = 4 * g8 + 2 * h_e + e;
        if (pi < 30) {
          const int cc = pi / 10, l = pi - 10 * cc;
          const float x = (cc == 0) ? p[0] : ((cc == 1) ? p[1] : p[2]);
          const float fl = __uint_as_float(kFreqPointBits[l]);
          float sn, cn;
          sincos_phase(x * fl, sn, cn);
          const float dgs = dgp[g8 >> 2][4 * (g8 & 3) + 2 * e], dgc = dgp[g8 >> 2][4 * (g8 & 3) + 2 * e + 1];
          const float contrib = fl * (cn * dgs - sn * dgc);
          if (cc == 0) dp[0] += contrib; else if (cc == 1) dp[1] += contrib; else dp[2] += contrib;
        }
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) dp[i] += __shfl_xor(dp[i], 32);
    if (valid_e && h_e == 0) {
      const float dtp = __builtin_fmaf(rf[RF_DWRD + 2], dp[2], __builtin_fmaf(rf[RF_DWRD + 1], dp[1], rf[RF_DWRD] * dp[0]));
      a.dt[m_e] += dtp;
    }
  }
}

// ------------------------------------------------------------------------------------------
// transposed weight image, two parts per step: fragment 2 s = hi, 2 s + 1 = mid of step s of the bf16 layout (bf16_weights.h bb_weight)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights_split_bwd(const Weights24 w, const float* __restrict__ fold, unsigned char* __restrict__ img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= BBF_NFRAG * 64) return;
  const int step = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
  u32x4 vh, vm;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
    const HiMidB e = split2b(bb_weight(w, fold, step, i, kk), bb_weight(w, fold, step, i, kk + 1));
    vh[q] = e.hi;
    vm[q] = e.mid;
  }
  unsigned char* dst = img + BF_BIAS_BYTES + (size_t)(2 * step) * BF_FRAG_BYTES + lane * 16;
  *reinterpret_cast<u32x4*>(dst) = vh;
  *reinterpret_cast<u32x4*>(dst + BF_FRAG_BYTES) = vm;
}

size_t split_bwd_image_bytes() { return (size_t)BF_BIAS_BYTES + (size_t)2 * BBF_NFRAG * BF_FRAG_BYTES; }

hipError_t launch_pack_weights_split_bwd(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  const int threads = BBF_NFRAG * 64;
  hipLaunchKernelGGL(k_pack_weights_split_bwd, dim3((threads + 255) / 256), dim3(256), 0, st, w, fold, img);
  return hipGetLastError();
}

hipError_t launch_field_bwd_split(const FieldBwdArgs& a, bool fine, hipStream_t st) {
  if (!a.wbf || !a.bmask || !a.bG || !a.bG2) return hipErrorInvalidValue;
  static std::atomic<unsigned long long> opted{0};
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_bwd_split<false>), reinterpret_cast<const void*>(&k_field_bwd_split<true>)},
                                        SPB_LDS_BYTES))
    return e;
  // (whole 256-sample groups of wave blocks, like the forward: every wave block the weight-gradient products read gets its -- zero -- gradients)
  const int wgs = ((a.M + SPB_WG / 2 - 1) / (SPB_WG / 2) + 1) / 2 * 2;
  if (fine)
    hipLaunchKernelGGL(k_field_bwd_split<true>, dim3(wgs), dim3(SPB_WG), SPB_LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL(k_field_bwd_split<false>, dim3(wgs), dim3(SPB_WG), SPB_LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace nerf
