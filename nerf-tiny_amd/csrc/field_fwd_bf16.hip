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
//     ReLU'd and rounded to bf16 pairwise (v_cvt_pk_bf16_f32 + v_pk_max_i16) and IS the next layer's B operand;
//     a layer's input and output live as 64 + 64 packed registers, so two waves fit on a SIMD and one wave's VALU work
//     (encoding, conversions, heads) runs in the shadow of the other's MFMAs;
//   * biases are fp32 accumulator start values read from LDS; sigma and colour heads are extra MFMA tiles (row 0 / rows
//     0..2 of a 32-row tile), 24 MFMAs instead of ~400 VALU instructions;
//   * point_info (no activation) is folded into dir_info's feature columns (bf16_common.h): 128 MFMAs per wave block less.
#include "bf16_stream.h"

#include <string.h>
#include "bf16_weights.h"
#include "ray_parts.h"

namespace nerf {

// ---- store schedule of the training variant (see bf16_stream.h): stores per tile epilogue ------------------------------
// The 16 (8) pieces of a layer's output and its 8 (4) mask words (one 16-byte store per lane) go out in ONE burst with the
// layer's last tile -- 17 KiB contiguous per wave instead of pieces spread over the layer (DRAM page locality of the writes).
struct FwdTiles { int s0, nft, ks, stores, last_extra; };
constexpr FwdTiles kFwdTiles[] = {
    {BFS_L0, 8, 4, 0, 17},        {BFS_L1, 8, 16, 0, 17},       {BFS_L1 + 128, 8, 16, 0, 17}, {BFS_L1 + 256, 8, 16, 0, 17},
    {BFS_L4, 8, 20, 0, 17},       {BFS_L5, 8, 16, 0, 17},       {BFS_L5 + 128, 8, 16, 0, 17}, {BFS_L5 + 256, 8, 16, 0, 17},
    {BFS_SIG, 1, 16, 0, 0},       {BFS_DIR, 4, 18, 0, 9},       {BFS_COL, 1, 8, 0, 0}};
constexpr BfStoreTable<BF_NFRAG> make_fwd_store_table() {
  BfStoreTable<BF_NFRAG> t{};
  int ev[BF_NFRAG + 64] = {};
  for (const FwdTiles& g : kFwdTiles)
    for (int f = 0; f < g.nft; ++f) ev[g.s0 + (f + 1) * g.ks + BF_EPI_POS] += g.stores + (f == g.nft - 1 ? g.last_extra : 0);
  int run = 0;
  for (int i = 0; i <= BF_NFRAG; ++i) {
    t.cum[i] = run;  // events strictly before step i
    run += ev[i];
  }
  return t;
}
constexpr BfStoreTable<BF_NFRAG> kFwdStoreTable = make_fwd_store_table();

constexpr int BFW_LDS_BYTES = BF_LDS_BYTES + BF_WG * 32;  // bias block + ring + 32 bytes per lane of parked direction encodings = 160 KiB
static_assert(BFW_LDS_BYTES <= 160 * 1024, "LDS of one CU");

template <bool SAVE>
struct FwdStream {
  static constexpr int NFRAG = BF_NFRAG, NCHUNK = BF_NCHUNK, NS = BF_NS, RING_OFF = BF_BIAS_BYTES, D = BF_D;
  static constexpr bool HAS_BIAS = true;
  static constexpr int PROLOGUE_STORES = SAVE ? 6 : 0;  // gamma_p (4 pieces) and gamma_d (2)
  __device__ static constexpr int stores_before(int idx) { return SAVE ? kFwdStoreTable.cum[idx] : 0; }
};

// A stream description for a 4-wave workgroup: the 16 pieces of a chunk over 4 waves (bf16_stream.h BfPW)
template <class Base> struct FourWaves : Base { static constexpr int PW = 4; };

// WAVES = 8: 256 samples per workgroup, two waves per SIMD.  WAVES = 4 (128 samples, one wave per SIMD) is what a SMALL pass gets: the
// coarse pass of a 512-ray batch is 128 workgroups of 256 samples -- half of the CUs idle -- or 256 of 128 (as field_fwd_bf16x.hip does).
template <bool SAVE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_field_fwd_bf16(const FieldArgs a, const FwdFuse fz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  BfCtx c;
  c.wimg = a.wbf;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = c.lane, j = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * (32 * WAVES) + c.wv * 32 + j;
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

  // ---- start the weight stream: bias block and chunks 0 .. BF_NS-2
  using S = std::conditional_t<WAVES == 4, FourWaves<FwdStream<SAVE>>, FwdStream<SAVE>>;
  bf_stream_start<S>(c);

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
#ifdef NERF_TIMING_NO_SINCOS  // (timing experiments only)
        sv = x; cv = x * 0.5f;
#else
        sincos_phase(x * __uint_as_float(kFreqPointBits[l]), sv, cv);
#endif
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
#ifdef NERF_TIMING_NO_SINCOS
        sv = x; cv = x * 0.5f;
#else
        sincos_phase(x * __uint_as_float(kFreqDirBits[l]), sv, cv);
#endif
      }
      gd[ks][q] = pack2(sv, cv);
    }

  // ---- training: this wave's block of the fragment-layout save buffers (bf16_common.h); every lane stores, also
  // lanes beyond the pass (they hold a copy of the last sample): the counted waits rely on the stores being issued
#ifdef NERF_TIMING_SAVE_ALIAS  // (timing experiments only: every save lands in the same few KiB -> the stores issue, HBM sees none)
  const int wb = c.wv;
#else
  const int wb = a.wb0 + blockIdx.x * WAVES + c.wv;
#endif
  unsigned mw[4];  // (every word is assigned by its even tile before the odd one ORs into it; words 2, 3 of a 4-tile layer are written as 0)
  // `lane16` = this lane's byte offset inside a piece.  The layers' epilogues pass a value produced AT their program point (mbcnt): left
  // alone the compiler forms every layer's 64-bit store address in the prologue and parks them -- and the lane pointer -- in scratch
  // (timing experiments only, results wrong -- DESIGN.md section 9: NERF_TIMING_SAVE_HALF writes every second piece of a saved tensor over its
  // neighbour, i.e. half the distinct bytes reach HBM -- what zero-compaction could at most take out of this kernel; NERF_TIMING_SAVE_SKIPALT
  // sends h1, h3, h5, h7 to one aliased KiB per wave -- what alternate-layer recompute in the weight-gradient kernels would not write)
  auto save_piece = [&](int tensor, int ks, const u32x4& v, unsigned lane16) {
#if defined(NERF_TIMING_SAVE_HALF)
    ks >>= 1;
#elif defined(NERF_TIMING_SAVE_SKIPALT)
    if (tensor == BS_H0 + 1 || tensor == BS_H0 + 3 || tensor == BS_H0 + 5 || tensor == BS_H0 + 7) {
      store_piece(a.bsave + (size_t)c.wv * BF_FRAG_BYTES + lane16, v);
      return;
    }
#endif
    store_piece(a.bsave + ((size_t)a.wb_tot * bs_cum(tensor) + (size_t)wb * bs_ks(tensor) + ks) * BF_FRAG_BYTES + lane16, v);
  };
  if (SAVE) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) save_piece(BS_GP, ks, gp[ks], lane * 16);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) save_piece(BS_GD, ks, gd[ks], lane * 16);
  }

  // the direction encodings are needed 976 fragments later: parked in the 16 KiB of LDS behind the ring (32 bytes per lane) instead of
  // 8 registers the allocator would spill to scratch (as in field_fwd_bf16x.hip)
  u32x4* const gd_park = reinterpret_cast<u32x4*>(lds + BF_LDS_BYTES) + 2 * threadIdx.x;
  gd_park[0] = gd[0];
  gd_park[1] = gd[1];

  // ---- bias block and chunk 0 have landed (mine), then everybody's
  u32x4 fr[S::D];
  bf_stream_first<S>(c, fr);

  u32x4 X[16], Y[16];
  f32x16 acc[2];
  acc[0] = bf_bias_tile(c, BFB_L0);
  // epilogue of a ReLU layer: tile f -> packed k-steps 2f, 2f+1 of the next layer's input; training: + save + alive mask
  auto relu_to = [&](u32x4* out, int tensor = -1, int mlayer = -1, int ntiles = 8) {
    return [&, out, tensor, mlayer, ntiles](int f, const f32x16& A) {
#ifdef NERF_TIMING_NO_EPI  // (timing experiments only: no conversion, half the values dropped)
      if (f >= 0) {
        for (int mh = 0; mh < 2; ++mh)
          for (int q = 0; q < 4; ++q) out[2 * f + mh][q] = __float_as_uint(A[8 * mh + 2 * q]) & 0x3f803f80u;
        return;
      }
#endif
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[2 * f + mh][q] = pack2_relu(A[8 * mh + 2 * q], A[8 * mh + 2 * q + 1]);
      if constexpr (SAVE) {
        // mask words of the layer's tiles, two per register: u16 [layer][wave block][lane][8 tiles]
        if (f & 1)
          mw[f >> 1] |= alive_bits(A) << 16;
        else
          mw[f >> 1] = alive_bits(A);
        if (f == ntiles - 1) {  // the layer's output is complete: one contiguous burst
          const unsigned lane16 = 16u * lane_id_here();
          u32x4 mv = {mw[0], mw[1], ntiles > 4 ? mw[2] : 0u, ntiles > 4 ? mw[3] : 0u};
          store_piece(reinterpret_cast<unsigned char*>(a.bmask) + ((size_t)mlayer * a.wb_tot + wb) * 1024 + lane16, mv);
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
            if (ks < 2 * ntiles) save_piece(tensor, ks, out[ks], lane16);
        }
      }
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](const f32x16& A) { epi(f, A); }; };
  auto nothing = [](const f32x16&) {};
  auto nothing_f = [](int, const f32x16&) {};

  // ---- layers 0..7 (nerf.py:104-112)
  bf_segment<S, BFS_L0, 8, 4, 0, BFB_L0, 0, BFB_L0 + 8>(c, fr, acc, gp, nullptr, relu_to(X, BS_H0 + 0, 0), nothing);
  bf_segment<S, BFS_L1, 8, 16, 0, BFB_L0 + 8, 0, BFB_L0 + 16>(c, fr, acc, X, nullptr, relu_to(Y, BS_H0 + 1, 1), last_of(relu_to(X, BS_H0 + 0, 0), 7));
  bf_segment<S, BFS_L1 + 128, 8, 16, 0, BFB_L0 + 16, 0, BFB_L0 + 24>(c, fr, acc, Y, nullptr, relu_to(X, BS_H0 + 2, 2), last_of(relu_to(Y, BS_H0 + 1, 1), 7));
  bf_segment<S, BFS_L1 + 256, 8, 16, 0, BFB_L0 + 24, 0, BFB_L0 + 32>(c, fr, acc, X, nullptr, relu_to(Y, BS_H0 + 3, 3), last_of(relu_to(X, BS_H0 + 2, 2), 7));
  bf_segment<S, BFS_L4, 8, 16, 4, BFB_L0 + 32, 0, BFB_L0 + 40>(c, fr, acc, Y, gp, relu_to(X, BS_H0 + 4, 4), last_of(relu_to(Y, BS_H0 + 3, 3), 7));
  bf_segment<S, BFS_L5, 8, 16, 0, BFB_L0 + 40, 0, BFB_L0 + 48>(c, fr, acc, X, nullptr, relu_to(Y, BS_H0 + 5, 5), last_of(relu_to(X, BS_H0 + 4, 4), 7));
  bf_segment<S, BFS_L5 + 128, 8, 16, 0, BFB_L0 + 48, 0, BFB_L0 + 56>(c, fr, acc, Y, nullptr, relu_to(X, BS_H0 + 6, 6), last_of(relu_to(Y, BS_H0 + 5, 5), 7));
  bf_segment<S, BFS_L5 + 256, 8, 16, 0, BFB_L0 + 56, 0, BFB_SIGMA>(c, fr, acc, X, nullptr, relu_to(Y, BS_H0 + 7, 7), last_of(relu_to(X, BS_H0 + 6, 6), 7));
  // ---- sigma head (one tile, row 0) on h7: sigma = |w_sigma . h7 + b|  (nerf.py:94, 113-115)
  float spre = 0.f;
  auto sig_epi = [&](const f32x16& A) { spre = A[0]; };
  bf_segment<S, BFS_SIG, 1, 16, 0, BFB_SIGMA, 0, BFB_DIR>(c, fr, acc, Y, nullptr, nothing_f, last_of(relu_to(Y, BS_H0 + 7, 7), 7));
  // ---- point_info folded into dir_info (bf16_common.h): c = relu(W_dir[:, :24] gamma_d + W_fold h7 + b_dir + W_dir[:, 24:] b_pi)
  // (nerf.py:117-118); its first tile also retires the sigma tile
  {
    const int lane_d = (int)lane_id_here();  // (re-derived: no address register kept across the stream)
    const u32x4* const back = reinterpret_cast<const u32x4*>(lds + BF_LDS_BYTES) + 2 * (c.wv * 64 + lane_d);
    gd[0] = back[0];
    gd[1] = back[1];
  }
  bf_segment<S, BFS_DIR, 4, 2, 16, BFB_DIR, 1, BFB_COL>(c, fr, acc, gd, Y, relu_to(X, BS_C, 8, 4), sig_epi);
  // (the sample index is re-derived from the lane id behind the stream -- mbcnt, not threadIdx: nothing to keep alive or spill)
  const int lane_e = (int)lane_id_here();
  const int m_e = blockIdx.x * (32 * WAVES) + c.wv * 32 + (lane_e & 31);
  const bool out_e = m_e < a.M && lane_e < 32;
  if (out_e) {
    a.sigma[m_e] = fabsf(spre);
    if (SAVE) a.spre[a.row0 + m_e] = spre;
  }
  // fused per-ray stage behind this pass (kernels.h FwdFuse): it takes the workgroup's sigma / rgb from LDS -- this wave's slot of the parked
  // direction encodings, read back just above -- so that it does not have to wait for the stores (and, with them, for every save of the
  // stream in front of them in the queue) to reach memory
  unsigned char* const park_w = lds + BF_LDS_BYTES + c.wv * 2048;
  if (SAVE && fz.mode != 0 && lane_e < 32) reinterpret_cast<float*>(park_w)[lane_e] = fabsf(spre);
  // ---- colour head: rows 0..2 of one tile, sigmoid (nerf.py:99, 119)
  bf_segment<S, BFS_COL, 1, 8, 0, BFB_COL, 1, -1>(c, fr, acc, X, nullptr, nothing_f, last_of(relu_to(X, BS_C, 8, 4), 3));
  {
    const float r0 = 1.0f / (1.0f + expf(-acc[1][0])), r1 = 1.0f / (1.0f + expf(-acc[1][1])), r2 = 1.0f / (1.0f + expf(-acc[1][2]));
    if (out_e) {
      a.rgb[(size_t)m_e * 3 + 0] = r0;
      a.rgb[(size_t)m_e * 3 + 1] = r1;
      a.rgb[(size_t)m_e * 3 + 2] = r2;
    }
    if (SAVE && fz.mode != 0 && lane_e < 32) {
      float* pr = reinterpret_cast<float*>(park_w + 128) + 3 * lane_e;
      pr[0] = r0; pr[1] = r1; pr[2] = r2;
    }
  }
  // ---- SMALL batches (kernels.h FwdFuse): the per-ray stage that would be the next launch runs HERE, on the workgroup's own rays -- its
  // samples are whole rays (32 WAVES = 2 or 4 coarse rays of 64, 1 or 2 fine rays of 128).  Behind a barrier: every wave's sigma / rgb
  // stores have completed (same CU: write-through L1, lines nobody has read in this launch) and the ring is free to be scratch.
  if constexpr (SAVE) {
    if (fz.mode != 0) {  // (kernel argument: uniform)
      // LDS only: the slots are written, nobody reads the ring any more -- NOT a wait for the stores of the stream (no vmcnt here)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      float* const scr = reinterpret_cast<float*>(lds + BF_BIAS_BYTES);
      const unsigned char* const park = lds + BF_LDS_BYTES;
      const int lane_f = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      auto wave_fence = [] {  // this wave's LDS writes before its LDS reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      };
      if (fz.mode == 1) {        // k_coarse: one ray per wave, WAVES / 2 rays
        if (c.wv < WAVES / 2) {
          float* w = scr + c.wv * (3 * 64 + 4 * 64);
          float* sg = w + 3 * 64;   // the ray's sigma [64] and rgb [64][3], gathered from the two waves' slots that hold them
          float* cl = sg + 64;
          sg[lane_f] = park_sigma(park, 64 * c.wv + lane_f);
#pragma unroll
          for (int k = 0; k < 3; ++k) cl[3 * lane_f + k] = park_rgb(park, 64 * c.wv + lane_f, k);
          wave_fence();
          const int ray_raw = blockIdx.x * (WAVES / 2) + c.wv;
          const size_t g0 = (size_t)(ray_raw < fz.c.B ? ray_raw : fz.c.B - 1) * 64;
          CoarseArgs ca = fz.c;
          ca.sigma = sg - g0;  // (the stage indexes [ray * Nc + i]: i = 0 lands on the scratch)
          ca.rgb = cl - 3 * g0;
          coarse_ray_stage(ca, ray_raw, lane_f, w, w + 64, w + 128, wave_fence);
        }
      } else {                   // k_merge<true>: WAVES / 4 rays.  Their 5 (WAVES / 4) channel sorts -- five independent sorts per ray (quirk
        // Q1) -- are dealt out over ALL waves, one 256-slot channel per job (the one-channel instance of k_merge's register network with
        // the original slots: the same compare-exchanges, the same permutation); then one wave per ray composites.
        constexpr int R = WAVES / 4;
        float* const val0 = scr;                                                   // [R][5][256] floats
        uint16_t* const idx0 = reinterpret_cast<uint16_t*>(scr + R * 5 * 256);    // [R][5][256] u16
        for (int j = c.wv; j < 5 * R; j += WAVES) {
          const int rl = j / 5, ch = j - 5 * rl;
          const int ray = blockIdx.x * R + rl;
          if (ray < fz.m.B) merge_channel_job(fz.m, ray, ch, lane_f, val0 + (rl * 5 + ch) * 256, idx0 + (rl * 5 + ch) * 256, park, 128 * rl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c.wv < R) {
          const int ray = blockIdx.x * R + c.wv;
          if (ray < fz.m.B) {
            constexpr int N = 192;
            const size_t gN = (size_t)ray * N;
            float cfin[3];
            merge_ray_composite<true>(val0 + c.wv * 5 * 256, idx0 + c.wv * 5 * 256, 256, N, fz.m.last, lane_f, fz.m.w ? fz.m.w + gN : nullptr,
                                      fz.m.bundle ? fz.m.bundle + gN * 5 : nullptr, fz.m.perm ? fz.m.perm + (size_t)ray * 5 * N : nullptr,
                                      fz.m.C_fine + (size_t)ray * 3, cfin);
            if (fz.C_true) {  // ray_loss's per-element work (inside nerf_hip_train_step): this ray's three elements
              if (lane_f < 3) {
                const size_t e = (size_t)ray * 3 + lane_f;
                const float cf = lane_f == 0 ? cfin[0] : (lane_f == 1 ? cfin[1] : cfin[2]);  // (C_fine from the registers, not read back)
                float d1, d2, term;
                ray_loss_element(fz.C_coarse[e], cf, fz.C_true[e], d1, d2, term);
                fz.dC_c[e] = d1;
                fz.dC_f[e] = d2;
                fz.loss_terms[e] = term;
              }
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight image
// ------------------------------------------------------------------------------------------
// first_thread = BF_NFRAG * 64: only the bias block (the 16x16x32 image shares it and brings its own fragments)
__global__ __launch_bounds__(256) void k_pack_weights_bf16(const Weights24 w, const float* __restrict__ fold, unsigned char* __restrict__ img, int first_thread) {
  const int gid = first_thread + blockIdx.x * 256 + threadIdx.x;
  if (gid < BF_NFRAG * 64) {
    const int frag = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
      v[q] = pack2(bf_weight(w, fold, frag, i, kk, h), bf_weight(w, fold, frag, i, kk + 1, h));
    }
    *reinterpret_cast<u32x4*>(img + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
  } else {
    const int b = gid - BF_NFRAG * 64;
    if (b < BF_BIAS_BYTES / 4) {
      const int tile = b >> 5, i = b & 31;
      reinterpret_cast<float*>(img)[b] = tile < BF_NBIAS_TILES ? bf_bias(w, fold, tile, i) : 0.f;
    }
  }
}

hipError_t launch_pack_weights_bf16(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  const int threads = BF_NFRAG * 64 + BF_BIAS_BYTES / 4;
  hipLaunchKernelGGL(k_pack_weights_bf16, dim3((threads + 255) / 256), dim3(256), 0, st, w, fold, img, 0);
  return hipGetLastError();
}

hipError_t launch_pack_bias_block_bf16(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_weights_bf16, dim3((BF_BIAS_BYTES / 4 + 255) / 256), dim3(256), 0, st, w, fold, img, BF_NFRAG * 64);
  return hipGetLastError();
}

hipError_t launch_field_fwd_bf16(const FieldArgs& a, bool save, hipStream_t st, const FwdFuse* fuse) {
  FwdFuse fz;
  if (fuse) fz = *fuse; else memset(&fz, 0, sizeof(fz));
  if (fz.mode && (!save || a.N != (fz.mode == 1 ? 64 : 128))) return hipErrorInvalidValue;  // whole rays per workgroup only at the shipped sample counts
  static std::atomic<unsigned long long> opted{0};  // >64 KiB of dynamic LDS needs an opt-in, once per device and kernel
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_fwd_bf16<false, 8>), reinterpret_cast<const void*>(&k_field_fwd_bf16<true, 8>),
                                                reinterpret_cast<const void*>(&k_field_fwd_bf16<false, 4>), reinterpret_cast<const void*>(&k_field_fwd_bf16<true, 4>)}, BFW_LDS_BYTES)) return e;
  const int wgs = (a.M + BF_WG / 2 - 1) / (BF_WG / 2);  // 256-sample workgroups: the pass's wave blocks are whole ones of these (api.hip wave_blocks)
  if (2 * wgs <= BF_SMALL_MAX_WGS && !bf16_four_waves_disabled()) {  // a small pass: 4-wave workgroups, so that every CU gets one
    if (save)
      hipLaunchKernelGGL((k_field_fwd_bf16<true, 4>), dim3(2 * wgs), dim3(256), BFW_LDS_BYTES, st, a, fz);
    else
      hipLaunchKernelGGL((k_field_fwd_bf16<false, 4>), dim3(2 * wgs), dim3(256), BFW_LDS_BYTES, st, a, fz);
    return hipGetLastError();
  }
  if (save)
    hipLaunchKernelGGL((k_field_fwd_bf16<true, 8>), dim3(wgs), dim3(BF_WG), BFW_LDS_BYTES, st, a, fz);
  else
    hipLaunchKernelGGL((k_field_fwd_bf16<false, 8>), dim3(wgs), dim3(BF_WG), BFW_LDS_BYTES, st, a, fz);
  return hipGetLastError();
}

}  // namespace nerf
