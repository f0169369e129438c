// field_fwd_ring.hip -- fp32 inference forward of the field query: k_field_fwd_reg's arithmetic (register-resident
// activations, one wave = 32 samples x all 256 features, v_mfma_f32_32x32x2_f32, bias-first accumulators, lazy ReLU) with
// the weight fragments shared through LDS instead of being fetched by every wave from L2 (MI355X / gfx950).
//
// What k_field_fwd_reg loses is ~140 cycles per k-block: 8 x 16-byte-per-lane global loads plus
// their address arithmetic in front of every 32 MFMAs (a vector-memory instruction costs ~16 cycles of fp32-MFMA issue).
// Here the 4 waves of a workgroup (one per SIMD, 128 samples) consume the same fragment stream (segment, k-block, tile --
// the order of use) in 16-KiB chunks (2 k-blocks) through a 4-slot LDS ring.  Each wave brings a quarter of a chunk:
// 4 plain 16-byte loads into registers (two chunk steps = ~3.7 us ahead of their use), 4 ds_write_b128 two chunks ahead of
// the readers, and takes its 8 fragments per k-block with 8 ds_read_b128: 2 vector-memory instructions per k-block instead
// of 8.  (Direct-to-LDS loads were measured first: an LDS-DMA instruction costs ~60 cycles of issue among MFMAs, which
// cancels the saving -- 593 vs 599 k rays/s.)  One s_barrier per chunk, at the top of the chunk's second k-block: it
// publishes the chunk written one step earlier and frees the slot of the chunk read one step earlier.
// Biases, the sigma row and the colour head live in a 16-KiB block at the front of LDS.  Same packed weight image as the
// other fp32 kernels.  Bit-identical results to k_field_fwd_reg.
#include "bf16_stream.h"  // glds16, wait_vmcnt, static_for

namespace nerf {

constexpr int RG_NS = 4, RG_CHUNK = 16;
constexpr int RG_NFRAG = 2304, RG_NCHUNK = RG_NFRAG / RG_CHUNK;  // 144
constexpr int RG_HEAD_BYTES = 16384;
constexpr int RG_LDS_BYTES = RG_HEAD_BYTES + RG_NS * RG_CHUNK * 1024;
constexpr int RG_WG = 256, RG_RM = 32;
// head block (floats): biases of layers 0..7, point_info bias, sigma row, colour matrix, sigma bias, colour bias
constexpr int RGH_BPI = 2048, RGH_WSIG = 2304, RGH_WCOL = 2560, RGH_BSIG = 2944, RGH_BCOL = 2945;

// stream segments in order of use: first fragment, segment id in the packed image, tiles per k-block
constexpr int kRgStart[12] = {0, 64, 320, 576, 832, 1088, 1152, 1408, 1664, 1920, 2176, 2304};
constexpr int kRgSeg[11] = {SEG_L0, SEG_L1, SEG_L2, SEG_L3, SEG_L4A, SEG_L4B, SEG_L5, SEG_L6, SEG_L7, SEG_PI, SEG_DIR};
__host__ __device__ constexpr int rg_seg_index(int frag) {
  int i = 0;
  while (i < 10 && frag >= kRgStart[i + 1]) ++i;
  return i;
}

struct RgCtx {
  const float4* wp;
  unsigned char* lds;
  unsigned ldl[3];              // LDS byte address lds + lane*16 + 64 KiB * {0, 1, 2}: a fragment read = one of these + a 16-bit immediate
  unsigned lds_base;
  int lane, wv;
  float4 stage[2][4];           // this wave's pieces of two chunks on their way from L2 to LDS
};

// this wave's four 1-KiB pieces of chunk C: piece p of the chunk is stream fragment 16C + p = (k-block, tile) of its segment;
// the packed image stores a segment tile-major.
template <int C>
__device__ __forceinline__ void rg_load_chunk(const RgCtx& c, float4 (&st)[4]) {
  constexpr int si = rg_seg_index(C * RG_CHUNK), seg = kRgSeg[si], nft = seg_nft(seg), kbn = seg_kb(seg);
  constexpr int local0 = C * RG_CHUNK - kRgStart[si];
  const float4* base = c.wp + seg_off4(seg) + c.lane;
  int wv = c.wv;
  asm volatile("" : "+s"(wv));  // the address arithmetic stays here (hoisted to the kernel top it floods the SGPR file)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned p = 4u * (unsigned)wv + e, sl = local0 + p;
    const unsigned kb = sl / (unsigned)nft, f = sl % (unsigned)nft;
    st[e] = base[(size_t)(f * kbn + kb) * 64];
  }
}
template <int C>
__device__ __forceinline__ void rg_write_chunk(const RgCtx& c, const float4 (&st)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float __attribute__((ext_vector_type(4))) f32x4;
  typedef f32x4 __attribute__((address_space(3)))* lds_f4_p;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned p = 4u * (unsigned)c.wv + e;
    const f32x4 v = {st[e].x, st[e].y, st[e].z, st[e].w};
    *(lds_f4_p)(uintptr_t)(c.lds_base + RG_HEAD_BYTES + ((C % RG_NS) * RG_CHUNK + p) * 1024 + c.lane * 16) = v;
  }
#endif
}

// at the top of the k-block that starts at fragment position 8 of chunk C
template <int C>
__device__ __forceinline__ void rg_sync(RgCtx& c) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my pieces of chunk C + 1 (written one step ago) are in LDS ...
  __builtin_amdgcn_s_barrier();                        // ... and so are everybody's; everybody is past chunk C - 1
  asm volatile("" ::: "memory");
  if constexpr (C + 2 < RG_NCHUNK) rg_write_chunk<C + 2>(c, c.stage[C & 1]);  // into the slot of chunk C - 2
  if constexpr (C + 4 < RG_NCHUNK) rg_load_chunk<C + 4>(c, c.stage[C & 1]);
}

__device__ __forceinline__ float4 rg_frag(const RgCtx& c, int pos) {
  const int off = RG_HEAD_BYTES + (((pos / RG_CHUNK) % RG_NS) * RG_CHUNK + pos % RG_CHUNK) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float __attribute__((ext_vector_type(4))) f32x4;
  typedef const f32x4 __attribute__((address_space(3)))* lds_f4_p;
  const f32x4 v = *(lds_f4_p)(uintptr_t)(c.ldl[off >> 16] + (off & 0xffff));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  (void)off;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}

struct RgStage { float4 w[8]; };

// ReLU as ONE integer max (see field_fwd_reg.hip)
__device__ __forceinline__ float rg_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ __forceinline__ float rg_f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// acc[f] (+)= sum_k W[f-tile][k] * act(prev)[k] over KB k-blocks of 8, fragments S0 .. S0 + KB*NFT of the stream.
// st0 holds k-block 0 on entry and the first NNFT fragments after the segment on exit.  HAS_BIAS (bias in LDS): one extra
// MFMA per tile starts the accumulator at the bias (same rounding order as ATen's addmm and as the other kernels).
template <int S0, int KB, int NFT, int NNFT, bool ZERO_INIT, bool RELU_IN, bool HAS_BIAS>
__device__ __forceinline__ void rg_layer(RgCtx& c, const f32x16* prev, f32x16* acc, RgStage& st0, const float* bias) {
  constexpr int KT = KB / 4;
  static_assert(KB % 2 == 0, "even number of k-blocks");
  const int lane = c.lane;
  RgStage st1;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_BIAS) {  // (not a null test: LDS offset 0 IS the first bias)
    const float one_h0 = (lane < 32) ? 1.0f : 0.0f;
#pragma unroll
    for (int f = 0; f < NFT; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias[f * 32 + (lane & 31)], one_h0, zero, 0, 0, 0);
  }
  f32x16 tin[2];
  auto activate = [&](int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // the copy is opaque and pinned to this program point: otherwise every tile's ReLU is hoisted to the top of the
      // layer, or merged with the sigma head's ReLU of the same accumulators (128 values live across a layer)
      float v = prev[t][r];
      asm volatile("" : "+v"(v));
      tin[t & 1][r] = RELU_IN ? rg_relu(v) : v;
    }
  };
  activate(0);
  static_for<KB>([&](auto KBI) {
    constexpr int kb = KBI, P = S0 + kb * NFT;
    RgStage& ld = (kb & 1) ? st0 : st1;
    const RgStage& cur = (kb & 1) ? st1 : st0;
    if constexpr (P % RG_CHUNK == 8) rg_sync<P / RG_CHUNK>(c);
    if constexpr (kb + 1 < KB) {
#pragma unroll
      for (int f = 0; f < NFT; ++f) ld.w[f] = rg_frag(c, P + NFT + f);
    } else if constexpr (NNFT > 0) {
#pragma unroll
      for (int f = 0; f < NNFT; ++f) ld.w[f] = rg_frag(c, S0 + KB * NFT + f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((kb & 3) == 2 && (kb >> 2) + 1 < KT) activate((kb >> 2) + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float b = tin[(kb >> 2) & 1][4 * (kb & 3) + s];
#pragma unroll
      for (int f = 0; f < NFT; ++f) {
        if (ZERO_INIT && !HAS_BIAS && kb == 0 && s == 0)
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg_f4c(cur.w[f], s), b, zero, 0, 0, 0);
        else
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg_f4c(cur.w[f], s), b, acc[f], 0, 0, 0);
      }
    }
  });
}

__global__ __launch_bounds__(RG_WG, 1) void k_field_fwd_ring(const FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  RgCtx c;
  c.wp = a.wp;
  c.lds = lds;
  c.lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  c.lane = threadIdx.x & 63;
  c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c.ldl[i] = c.lds_base + c.lane * 16 + i * 65536;
    asm volatile("" : "+v"(c.ldl[i]));
  }
  const int lane = c.lane, j = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * (RG_WG / 2) + c.wv * RG_RM + j;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int ray = mc / a.N;
  const float* rf = a.rayf + (size_t)ray * RAYF;

  // ---- ordinary loads first (drained before the first direct-to-LDS load is issued)
  float p[3];
  sample_point(rf, a.t[mc], p);
#pragma unroll
  for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(p[i]));

  // ---- head block (4 pieces per wave; it sits behind the packed segments in the image) and chunks 0, 1 on their way
  float4 headp[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) headp[e] = a.wp[PACKED_ALL_F4 + (4 * c.wv + e) * 64 + lane];
  rg_load_chunk<0>(c, c.stage[0]);
  rg_load_chunk<1>(c, c.stage[1]);

  // ---- sample encoding straight into B-operand registers: gp[t][4g + s] = gamma_p[k], k = 32t + 8g + 4h + s
  f32x16 gp[2];
#pragma unroll
  for (int g8 = 0; g8 < 8; ++g8) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pi = 4 * g8 + 2 * h + e;  // (sin, cos) pair index: k = 2 pi
      float sv = 0.f, cv = 0.f;
      if (pi < 30) {
        const int cc = pi / 10, l = pi - 10 * cc;
        const float x = (cc == 0) ? p[0] : ((cc == 1) ? p[1] : p[2]);
        sincos_phase(x * __uint_as_float(kFreqPointBits[l]), sv, cv);
      }
      gp[g8 >> 2][4 * (g8 & 3) + 2 * e] = sv;
      gp[g8 >> 2][4 * (g8 & 3) + 2 * e + 1] = cv;
    }
  }

  // ---- head block and chunks 0, 1 into LDS (mine, then everybody's); chunks 2, 3 on their way
  {
    typedef float __attribute__((ext_vector_type(4))) f32x4;
    typedef f32x4 __attribute__((address_space(3)))* lds_f4_p;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 v = {headp[e].x, headp[e].y, headp[e].z, headp[e].w};
      *(lds_f4_p)(uintptr_t)(c.lds_base + (4 * c.wv + e) * 1024 + lane * 16) = v;
    }
  }
  rg_write_chunk<0>(c, c.stage[0]);
  rg_write_chunk<1>(c, c.stage[1]);
  asm volatile("" ::: "memory");
  rg_load_chunk<2>(c, c.stage[0]);
  rg_load_chunk<3>(c, c.stage[1]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const float* const head = reinterpret_cast<const float*>(lds);
  RgStage st0;
#pragma unroll
  for (int f = 0; f < 8; ++f) st0.w[f] = rg_frag(c, f);

  f32x16 A[8], B[8];
  // ---- layers 0..7 (nerf.py:104-112; layer 4 = cat(h3, gamma_p), hidden first)
  rg_layer<0, 8, 8, 8, true, false, true>(c, gp, A, st0, head + 0 * 256);
  rg_layer<64, 32, 8, 8, true, true, true>(c, A, B, st0, head + 1 * 256);
  rg_layer<320, 32, 8, 8, true, true, true>(c, B, A, st0, head + 2 * 256);
  rg_layer<576, 32, 8, 8, true, true, true>(c, A, B, st0, head + 3 * 256);
  rg_layer<832, 32, 8, 8, true, true, true>(c, B, A, st0, head + 4 * 256);
  rg_layer<1088, 8, 8, 8, false, false, false>(c, gp, A, st0, nullptr);
  rg_layer<1152, 32, 8, 8, true, true, true>(c, A, B, st0, head + 5 * 256);
  rg_layer<1408, 32, 8, 8, true, true, true>(c, B, A, st0, head + 6 * 256);
  rg_layer<1664, 32, 8, 8, true, true, true>(c, A, B, st0, head + 7 * 256);
  // ---- sigma head on h7 = relu(B) (VALU): sigma = |w_sigma . h7 + b|  (nerf.py:94, 115)
  {
    asm volatile("" ::: "memory");  // the 64 LDS reads below stay below layer 7 (hoisted, they cost 256 registers)
    const float* ws = head + RGH_WSIG + 4 * h;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      asm volatile("" ::: "memory");  // one tile's weights at a time
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 wq = *reinterpret_cast<const float4*>(ws + 32 * t + 8 * g);
        s = __builtin_fmaf(rg_relu(B[t][4 * g + 0]), wq.x, s);
        s = __builtin_fmaf(rg_relu(B[t][4 * g + 1]), wq.y, s);
        s = __builtin_fmaf(rg_relu(B[t][4 * g + 2]), wq.z, s);
        s = __builtin_fmaf(rg_relu(B[t][4 * g + 3]), wq.w, s);
      }
    }
    s += __shfl_xor(s, 32);
    if (valid && h == 0) a.sigma[m] = fabsf(s + head[RGH_BSIG]);
  }
  // ---- point_info: 256 -> 256, no activation; next segment = dir_info (4 tiles)
  rg_layer<1920, 32, 8, 4, true, true, true>(c, B, A, st0, head + RGH_BPI);
  // ---- dir_info: cat(gamma_d, feat) -> 128, ReLU.  The feature part runs here from zero; the direction part + bias
  // (dvec, per ray) is added in the colour head below.  (Pre-loading the accumulators with dvec, as k_field_fwd_reg does,
  // makes this kernel's register allocation collapse: ~450 spills.)
  rg_layer<2176, 32, 4, 0, true, false, false>(c, A, B, st0, nullptr);
  // ---- colour head (VALU): rgb = sigmoid(W_c relu(.) + b)  (nerf.py:99, 119)
  {
    asm volatile("" ::: "memory");  // likewise: the colour matrix is read after dir_info, not during it
    const float* wc = head + RGH_WCOL + 4 * h;
    const float* dv = a.dvec + (size_t)ray * HALF + 4 * h;
    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q0 = *reinterpret_cast<const float4*>(wc + 32 * t + 8 * g);
        const float4 q1 = *reinterpret_cast<const float4*>(wc + HALF + 32 * t + 8 * g);
        const float4 q2 = *reinterpret_cast<const float4*>(wc + 2 * HALF + 32 * t + 8 * g);
        const float4 dq = *reinterpret_cast<const float4*>(dv + 32 * t + 8 * g);
        const float c0 = rg_relu(B[t][4 * g + 0] + dq.x), c1 = rg_relu(B[t][4 * g + 1] + dq.y);
        const float c2 = rg_relu(B[t][4 * g + 2] + dq.z), c3 = rg_relu(B[t][4 * g + 3] + dq.w);
        z0 = __builtin_fmaf(c3, q0.w, __builtin_fmaf(c2, q0.z, __builtin_fmaf(c1, q0.y, __builtin_fmaf(c0, q0.x, z0))));
        z1 = __builtin_fmaf(c3, q1.w, __builtin_fmaf(c2, q1.z, __builtin_fmaf(c1, q1.y, __builtin_fmaf(c0, q1.x, z1))));
        z2 = __builtin_fmaf(c3, q2.w, __builtin_fmaf(c2, q2.z, __builtin_fmaf(c1, q2.y, __builtin_fmaf(c0, q2.x, z2))));
      }
    }
    z0 += __shfl_xor(z0, 32);
    z1 += __shfl_xor(z1, 32);
    z2 += __shfl_xor(z2, 32);
    if (valid && h == 0) {
      a.rgb[(size_t)m * 3 + 0] = 1.0f / (1.0f + expf(-(z0 + head[RGH_BCOL + 0])));
      a.rgb[(size_t)m * 3 + 1] = 1.0f / (1.0f + expf(-(z1 + head[RGH_BCOL + 1])));
      a.rgb[(size_t)m * 3 + 2] = 1.0f / (1.0f + expf(-(z2 + head[RGH_BCOL + 2])));
    }
  }
}

// head block behind the packed segments: [8][256] layer biases, point_info bias, sigma row, colour matrix, sigma / colour bias
__global__ __launch_bounds__(256) void k_pack_head_block(const Weights24 w, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= RG_HEAD_BYTES / 4) return;
  float v = 0.f;
  if (i < 2048) v = w.p[2 * (i >> 8) + 1][i & 255];
  else if (i < RGH_WSIG) v = w.p[B_PI][i - RGH_BPI];
  else if (i < RGH_WCOL) v = w.p[W_SIGMA][i - RGH_WSIG];
  else if (i < RGH_BSIG) v = w.p[W_COLOR][i - RGH_WCOL];
  else if (i == RGH_BSIG) v = w.p[B_SIGMA][0];
  else if (i < RGH_BCOL + 3) v = w.p[B_COLOR][i - RGH_BCOL];
  out[i] = v;
}

hipError_t launch_pack_head_block(const Weights24& w, float4* packed, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_head_block, dim3(RG_HEAD_BYTES / 4 / 256), dim3(256), 0, st, w, reinterpret_cast<float*>(packed + PACKED_ALL_F4));
  return hipGetLastError();
}

hipError_t launch_field_fwd_ring(const FieldArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> opted{0};  // >64 KiB of dynamic LDS needs an opt-in, once per device and kernel
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_field_fwd_ring)}, RG_LDS_BYTES)) return e;
  const int wgs = (a.M + RG_WG / 2 - 1) / (RG_WG / 2);
  hipLaunchKernelGGL(k_field_fwd_ring, dim3(wgs), dim3(RG_WG), RG_LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace nerf
