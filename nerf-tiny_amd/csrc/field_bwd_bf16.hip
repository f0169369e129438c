// field_bwd_bf16.hip -- backward dX chain of the bf16-MLP field query (training half of BASELINE.json cfg3), MI355X / gfx950.
//
// Mirror image of field_fwd_bf16.hip on the same machinery (bf16_stream.h): a wave owns 32 samples, the stream now holds
// the TRANSPOSED weights from the colour head back to layer 0, and the fp32 accumulator of "d input" of one layer,
// masked with the layer's saved ReLU bits and rounded to bf16, is the B operand of the next (earlier) layer:
//     dz -> dc -> dpre_dir -> dh7 (W_fold^T: point_info folded into dir_info; + w_sigma dspre) -> dpre7 -> ... -> dpre0 [-> d gamma_p -> dt, fine pass]
// Every masked accumulator (= pre-activation gradient, the A operand of that layer's dW GEMM in dw_bf16.hip) is also
// written to the gradient buffer in fragment layout (bf16_common.h), 1 KiB per store instruction.
// The wave's ReLU masks (9 layers x 8 tiles x 64 lanes x u16 = 9 KiB) are brought to LDS once, by the same direct-to-LDS
// loads as the weights and ahead of them, so that no ordinary load sits in the counted vmcnt queue of the stream; for
// that the ring shrinks to 5 slots (3.5 chunks of look-ahead).  LDS: 72 KiB masks + 80 KiB ring.
// Gradients are rounded to bf16 where they enter an MFMA (standard mixed precision); sums stay fp32.
#include "bf16_stream.h"
#include "bf16_weights.h"
#include "ray_parts_bwd.h"
#include "ray_parts.h"

#include <string.h>

namespace nerf {

#ifndef NERF_BB_NS  // (timing experiments only: ring depth of the chain's stream)
#define NERF_BB_NS 5
#endif
constexpr int BB_NS = NERF_BB_NS;
constexpr int BB_MASK_BYTES = 8 * BM_LAYERS * 1024;  // per workgroup: wave w, layer l at (w * 9 + l) * 1024
constexpr int BB_LDS_BYTES = BB_MASK_BYTES + BB_NS * BF_CHUNK * BF_FRAG_BYTES;

struct BwdTiles { int s0, nft, ks, last_stores; };  // a layer's gradient pieces go out in one burst with its last tile

template <bool FINE>
struct BwdStream {
  static constexpr int NFRAG = FINE ? BBF_NFRAG : BBC_NFRAG;
  static constexpr int NCHUNK = FINE ? BBF_NCHUNK : BBC_NCHUNK;
  static constexpr int NS = BB_NS, RING_OFF = BB_MASK_BYTES, D = 4;
  static constexpr bool HAS_BIAS = false;
  static constexpr int PROLOGUE_STORES = 2;  // the dz / dspre fragment and its zero partner
  static constexpr int L3T = BBS_L3T;
  static constexpr BfStoreTable<NFRAG> make() {
    const BwdTiles tiles[] = {{BBS_COLT, 4, 4, 8},        {BBS_FOLDT, 8, 9, 16},      {BBS_L7T, 8, 16, 16},
                              {BBS_L7T + 128, 8, 16, 16}, {BBS_L7T + 256, 8, 16, 16}, {BBS_L4T, 8, 16, 16},     {L3T, 8, 16, 16},
                              {L3T + 128, 8, 16, 16},     {L3T + 256, 8, 16, 16}};  // the d gamma_p tiles store nothing
    BfStoreTable<NFRAG> t{};
    int ev[NFRAG + 64] = {};
    for (const BwdTiles& g : tiles)
      {
        int e = g.s0 + g.nft * g.ks + BF_EPI_POS;  // epilogue of the segment's last tile
        // (an epilogue deferred past the end of the coarse stream runs after the last sync point)
        ev[e < NFRAG + 64 ? e : NFRAG + 63] += g.last_stores;
      }
    int run = 0;
    for (int i = 0; i <= NFRAG; ++i) {
      t.cum[i] = run;
      run += ev[i];
    }
    return t;
  }
  static constexpr BfStoreTable<NFRAG> tab = make();
  __device__ static constexpr int stores_before(int idx) { return tab.cum[idx]; }
};

template <class Base> struct FourWavesB : Base { static constexpr int PW = 4; };

// WAVES = 4 (128 samples per workgroup, one wave per SIMD): small passes, so that every CU gets a workgroup (field_fwd_bf16.hip)
template <bool FINE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_field_bwd_bf16(const FieldBwdArgs a, const BwdFuse fz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using S = std::conditional_t<WAVES == 4, FourWavesB<BwdStream<FINE>>, BwdStream<FINE>>;
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
#ifdef NERF_TIMING_SAVE_ALIAS  // (timing experiments only: masks read from / gradients written to the same few KiB)
  const int wb = c.wv;
#else
  const int wb = a.wb0 + blockIdx.x * WAVES + c.wv;
#endif

  // ---- SMALL batches (kernels.h BwdFuse): the per-ray backward stage that would be the launch in FRONT of this one runs here first, on the
  // workgroup's own rays (32 WAVES samples = 1 or 2 fine rays, 2 or 4 coarse rays); its outputs (d rgb, d sigma, d t of the samples) go to
  // the same global buffers and are read back below, behind a barrier (stores complete; same CU, lines nobody has read in this launch)
  if (fz.mode != 0) {  // (kernel argument: uniform)
    float* const scr = reinterpret_cast<float*>(lds);  // nothing has been brought to LDS yet
    auto wave_fence = [] {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    if (FINE) {
      if (c.wv < WAVES / 4) {
        const int ray = blockIdx.x * (WAVES / 4) + c.wv;
        if (ray < fz.m.B) merge_bwd_ray(fz.m, ray, lane, scr + c.wv * 4 * 192, wave_fence);
      }
      // the loss VALUE (nobody on the device waits for it): the 3 B summands the forward's epilogues left, added in k_ray_loss's order
      if (blockIdx.x == 0 && fz.loss_terms) ray_loss_sum<64 * WAVES>(fz.loss_terms, 3 * fz.m.B, fz.loss, scr + 4 * 4 * 192, threadIdx.x);
    } else {
      if (c.wv < WAVES / 2) {
        const int ray_raw = blockIdx.x * (WAVES / 2) + c.wv;
        const bool live = ray_raw < fz.c.B;
        float* w = scr + c.wv * (5 * 64 + 64);
        coarse_bwd_ray(fz.c, live ? ray_raw : fz.c.B - 1, live, lane, w, reinterpret_cast<uint16_t*>(w + 5 * 64), wave_fence);
      }
    }
    __syncthreads();
  }
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
  // as a B operand / gradient fragment: inputs 0..2 = dz, input 3 = dspre  (k = 4h + s: lane half 0, slots 0..3)
  u32x4 zin[4];
  zin[0][0] = h == 0 ? pack2(dz[0], dz[1]) : 0u;
  zin[0][1] = h == 0 ? pack2(dz[2], ds) : 0u;
  zin[0][2] = 0u;
  zin[0][3] = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(zin[0][q]));  // the loads above are complete from here on
  zin[1] = zin[2] = zin[3] = u32x4{0u, 0u, 0u, 0u};

  // ---- this wave's ReLU masks -> LDS (9 x 1 KiB), then the weight stream
#pragma unroll
  for (int l = 0; l < BM_LAYERS; ++l)
    glds16(reinterpret_cast<const unsigned char*>(a.bmask) + ((size_t)l * a.wb_tot + wb) * 1024 + lane * 16,
           c.lds_base + (c.wv * BM_LAYERS + l) * 1024);
  bf_stream_start<S>(c);

  // `lane16` = this lane's byte offset inside a piece; the epilogues pass one produced at THEIR program point (lane_id_here): an
  // address register kept from the prologue lives across the whole stream -- the allocator parked five of them in scratch
  auto grad_piece = [&](int tensor, int ks, const u32x4& v, unsigned lane16) {
#ifdef NERF_TIMING_G_HALF  // (timing experiments only, results wrong: half the distinct gradient bytes reach HBM)
    ks >>= 1;
#endif
    store_piece(a.bG + ((size_t)a.wb_tot * bg_cum(tensor) + (size_t)wb * bg_ks(tensor) + ks) * BF_FRAG_BYTES + lane16, v);
  };
  grad_piece(BG_Z, 0, zin[0], lane * 16);
  grad_piece(BG_Z, 1, zin[1], lane * 16);

  u32x4 fr[S::D];
  bf_stream_first<S>(c, fr);

  u32x4 X[16], Y[16];
  f32x16 acc[2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc[0] = zero;
  const uint16_t* const mk = reinterpret_cast<const uint16_t*>(lds + c.wv * BM_LAYERS * 1024) + lane * 8;  // [layer][lane][8 tiles]
  // epilogue: d(input) tile f -> (mask with the ReLU bits of `mlayer`) -> packed k-steps 2f, 2f+1 of the next GEMM + store
  auto grad_to = [&](u32x4* out, int tensor, int mlayer, int ntiles = 8) {
    return [&, out, tensor, mlayer, ntiles](int f, const f32x16& A) {
      f32x16 D = A;
      if (mlayer >= 0) {
        const int bits = mk[mlayer * 512 + f];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          D[r] = __uint_as_float(__float_as_uint(A[r]) & (unsigned)__builtin_amdgcn_sbfe(bits, 15 - r, 1));
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[2 * f + mh][q] = pack2(D[8 * mh + 2 * q], D[8 * mh + 2 * q + 1]);
      if (f == ntiles - 1) {  // the whole gradient tensor of this wave block in one contiguous burst
        const unsigned lane16 = 16u * lane_id_here();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          if (ks < 2 * ntiles) grad_piece(tensor, ks, out[ks], lane16);
      }
    };
  };
  auto last_of = [](auto epi, int f) { return [epi, f](const f32x16& A) { epi(f, A); }; };
  auto nothing = [](const f32x16&) {};

  // colour head: dc = W_color^T dz, through c's ReLU -> dpre_dir
  bf_segment<S, BBS_COLT, 4, 4, 0, -1, 0, -1>(c, fr, acc, zin, nullptr, grad_to(X, BG_D, 8, 4), nothing);
  // dir_info and point_info as ONE transposed layer (point_info has no activation: folded, bf16_common.h) + the sigma head:
  // dh7 = W_fold^T dpre_dir + w_sigma dspre, through h7's ReLU
  bf_segment<S, BBS_FOLDT, 8, 8, 1, -1, 0, -1>(c, fr, acc, X, zin, grad_to(Y, BG_L0 + 7, 7), last_of(grad_to(X, BG_D, 8, 4), 3));
  bf_segment<S, BBS_L7T, 8, 16, 0, -1, 0, -1>(c, fr, acc, Y, nullptr, grad_to(X, BG_L0 + 6, 6), last_of(grad_to(Y, BG_L0 + 7, 7), 7));
  bf_segment<S, BBS_L7T + 128, 8, 16, 0, -1, 0, -1>(c, fr, acc, X, nullptr, grad_to(Y, BG_L0 + 5, 5), last_of(grad_to(X, BG_L0 + 6, 6), 7));
  bf_segment<S, BBS_L7T + 256, 8, 16, 0, -1, 0, -1>(c, fr, acc, Y, nullptr, grad_to(X, BG_L0 + 4, 4), last_of(grad_to(Y, BG_L0 + 5, 5), 7));
  bf_segment<S, BBS_L4T, 8, 16, 0, -1, 0, -1>(c, fr, acc, X, nullptr, grad_to(Y, BG_L0 + 3, 3), last_of(grad_to(X, BG_L0 + 4, 4), 7));
  bf_segment<S, BBS_L3T, 8, 16, 0, -1, 0, -1>(c, fr, acc, Y, nullptr, grad_to(X, BG_L0 + 2, 2), last_of(grad_to(Y, BG_L0 + 3, 3), 7));
  bf_segment<S, BBS_L3T + 128, 8, 16, 0, -1, 0, -1>(c, fr, acc, X, nullptr, grad_to(Y, BG_L0 + 1, 1), last_of(grad_to(X, BG_L0 + 2, 2), 7));
  bf_segment<S, BBS_L3T + 256, 8, 16, 0, -1, 0, -1>(c, fr, acc, Y, nullptr, grad_to(X, BG_L0 + 0, 0), last_of(grad_to(Y, BG_L0 + 1, 1), 7));
  if constexpr (!FINE) {
    grad_to(X, BG_L0 + 0, 0)(7, acc[1]);  // the last tile of the stream
  } else {
    // ---- d gamma_p (fp32) = W_0^T dpre0 + W_4[:, 256:]^T dpre4  (nerf.py:104, 109).  dpre4 is long gone from the
    // registers: the wave reads back the 16 pieces it stored itself ~500 MFMAs ago (complete: every counted wait since
    // has retired them; nobody on this CU has read those lines, so no stale copy can be hit), while G0T runs.
    // One segment: tile f of d gamma_p runs over the 16 k-steps of dpre0 (X, W_0^T) and then the 16 k-steps of dpre4 (Y, W_4[:, 256:]^T)
    // in the same accumulator (the weight image interleaves the two matrices per tile): no second accumulator pair to add up.
    f32x16 dgp[2];
    {
      const unsigned lane16 = 16u * lane_id_here();
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        Y[ks] = *reinterpret_cast<const u32x4*>(a.bG + ((size_t)a.wb_tot * bg_cum(BG_L0 + 4) + (size_t)wb * 16 + ks) * BF_FRAG_BYTES + lane16);
    }
    bf_segment<S, BBS_G0T, 2, 16, 16, -1, 0, -1>(c, fr, acc, X, Y, [&](int, const f32x16& A) { dgp[0] = A; },
                                                 last_of(grad_to(X, BG_L0 + 0, 0), 7));
    dgp[1] = acc[1];
    // gamma -> point -> depth (t_fine is not detached, quirk Q9).  dgp[t][4g + 2e], [.. + 1] = d loss / d (sin, cos) of
    // pair pi = 16t + 4g + 2h + e
    // (the sample index is re-derived from the lane id HERE: nothing of the prologue stays alive across the stream, and the geometry
    // loads and the 16 sincos below cannot be hoisted over it)
    const int lane_e = (int)lane_id_here();
    const int h_e = lane_e >> 5;
    const int m_e = blockIdx.x * (32 * WAVES) + c.wv * 32 + (lane_e & 31);
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
// transposed weight image: fragment (f, ks), lane (i, h), slot s  =  W[out = 16ks + 4h + (s&3) + 8(s>>2)][in = 32f + i]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights_bf16_bwd(const Weights24 w, const float* __restrict__ fold, unsigned char* __restrict__ img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= BBF_NCHUNK * BF_CHUNK * 64) return;
  const int frag = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
  u32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
    v[q] = pack2(bb_weight(w, fold, frag, i, kk), bb_weight(w, fold, frag, i, kk + 1));
  }
  *reinterpret_cast<u32x4*>(img + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
}

hipError_t launch_pack_weights_bf16_bwd(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st) {
  const int threads = BBF_NCHUNK * BF_CHUNK * 64;
  hipLaunchKernelGGL(k_pack_weights_bf16_bwd, dim3((threads + 255) / 256), dim3(256), 0, st, w, fold, img);
  return hipGetLastError();
}

hipError_t launch_field_bwd_bf16(const FieldBwdArgs& a, bool fine, hipStream_t st, const BwdFuse* fuse) {
  BwdFuse fz;
  if (fuse) fz = *fuse; else memset(&fz, 0, sizeof(fz));
  if (fz.mode && (a.N != (fine ? 128 : 64) || (fz.mode == 1) != fine)) return hipErrorInvalidValue;  // whole rays per workgroup only at the shipped sample counts
  static std::atomic<unsigned long long> opted{0};
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_bwd_bf16<false, 8>), reinterpret_cast<const void*>(&k_field_bwd_bf16<true, 8>),
                                                reinterpret_cast<const void*>(&k_field_bwd_bf16<false, 4>), reinterpret_cast<const void*>(&k_field_bwd_bf16<true, 4>)}, BB_LDS_BYTES)) return e;
  const int wgs = (a.M + BF_WG / 2 - 1) / (BF_WG / 2);
  if (2 * wgs <= BF_SMALL_MAX_WGS && !bf16_four_waves_disabled()) {  // a small pass: 4-wave workgroups, so that every CU gets one
    if (fine)
      hipLaunchKernelGGL((k_field_bwd_bf16<true, 4>), dim3(2 * wgs), dim3(256), BB_LDS_BYTES, st, a, fz);
    else
      hipLaunchKernelGGL((k_field_bwd_bf16<false, 4>), dim3(2 * wgs), dim3(256), BB_LDS_BYTES, st, a, fz);
    return hipGetLastError();
  }
  if (fine)
    hipLaunchKernelGGL((k_field_bwd_bf16<true, 8>), dim3(wgs), dim3(BF_WG), BB_LDS_BYTES, st, a, fz);
  else
    hipLaunchKernelGGL((k_field_bwd_bf16<false, 8>), dim3(wgs), dim3(BF_WG), BB_LDS_BYTES, st, a, fz);
  return hipGetLastError();
}

}  // namespace nerf
