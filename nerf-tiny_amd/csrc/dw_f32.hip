// dw_f32.hip -- the weight-gradient GEMMs of the fp32 train step for MI355X (gfx950).
//
//  k_dw<NCA>      dW[out][in] = sum_m G[m][out] * X[m][in] as a split-M fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32 with the
//                 SAMPLE as the k index): both operands are read straight from their row-major HBM images (lane (q, h) <-
//                 columns 4q..4q+3 of G and 2q..2q+1 of X, row m + h), one 128 x 64 output block = 128 accumulators per
//                 wave, 8 waves (two per SIMD), one workgroup per CU, a branch-free 3-stage register rotation with pinned
//                 prefetches, per-wave partial slabs.  fp32 MFMA runs on the SIMD's fp32 lanes, so every VALU instruction
//                 in the loop is MFMA time lost: addresses are SCALAR (wave-uniform row cursor in SGPRs + one constant
//                 32-bit lane offset; the loop has no vector address arithmetic), and the bias gradients (column sums of
//                 G, four adds per k-step) ride on one of the waves that read the same G columns.
//                 NCA = 1: the thin heads as one product -- A = the [rows][4] buffer (dz_r, dz_g, dz_b, dsigma_pre), one
//                 32-row output tile per wave, X = [h7 | c]: rows 0..2 x c give the colour head, row 3 x h7 the sigma head,
//                 the column sums of A their biases.
//  k_dw_reduce    ONE launch per step: sums the slabs of all products in a fixed order (deterministic, no float atomics)
//                 and scatters into the nn.Linear-layout gradients.
//
// Autograd spans replaced: the weight / bias gradients of Network.forward (nerf.py:101-124) as produced by
// loss.backward() at nerf.py:473.
#include "field_common.h"

namespace nerf {

namespace {

constexpr int DW_UNROLL = 4;   // k-steps (row pairs) per pipeline stage
constexpr int DW_STAGES = 3;   // register stages in flight (2 prefetched ahead of the one being multiplied)
constexpr int DW_WAVES = 8;    // 512 threads, two waves per SIMD
constexpr int DW_ROWS = 2 * DW_UNROLL;  // rows per stage

__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
__device__ __forceinline__ float comp(const float2& v, int c) { return c == 0 ? v.x : v.y; }

template <int NCA>
struct DwStage {
  float4 a[DW_UNROLL];  // NCA == 1: only .x is used
  float2 b[DW_UNROLL];
};

// 16-byte / 8-byte loads in the SADDR form `scalar 64-bit base + 32-bit lane offset + immediate`: no vector address
// arithmetic.  Written as inline asm because the compiler re-associates base + row cursor + lane offset into per-lane
// 64-bit pointers and then advances those with vector adds every stage.  The compiler does not see these as memory
// operations, so the loop below counts its own `s_waitcnt vmcnt` (always 16: two younger stages of 8 loads) and threads the
// loaded registers through the wait so that no use can be scheduled above it.
typedef float f32x4 __attribute__((ext_vector_type(4)));  // register tuples for asm operands (HIP's float4 is a struct)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int IMM>
__device__ __forceinline__ float4 gload_x4(unsigned voff, const char* sbase) {
  f32x4 d;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  return make_float4(d[0], d[1], d[2], d[3]);
}
template <int IMM>
__device__ __forceinline__ float2 gload_x2(unsigned voff, const char* sbase) {
  f32x2 d;
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  return make_float2(d[0], d[1]);
}

// Main loop over rows [r_begin, r_end) (wave-uniform).  CHECK = false: every row of every stage is in range; with ASM_LOADS
// the row cursor, its clamp and the bases are scalar-ALU work and the loop holds no vector instruction but MFMAs (and, BIAS,
// the four column-sum adds per k-step).  CHECK = true (the ragged tail of a pass only): rows are clamped per lane and rows
// past the end contribute nothing.  Rows are 32-bit (a pass has < 2^31 rows and < 4 GiB per operand).
template <int NCA, bool CHECK, bool BIAS>
__device__ __forceinline__ void dw_rows(const char* __restrict__ gbase, const char* __restrict__ xbase, unsigned ga_row_bytes,
                                        unsigned voff_a, unsigned voff_b, int r_begin, int r_end, int h, bool a_live,
                                        f32x16 (&acc)[NCA][2], float (&bsum)[4]) {
#ifdef DW_NO_ASM_LOADS  // A/B switch: plain loads (the compiler then advances per-lane 64-bit pointers with vector adds)
  constexpr bool ASM_LOADS = false;
#else
  constexpr bool ASM_LOADS = (NCA == 4) && !CHECK;
#endif
  auto load = [&](int r0, DwStage<NCA>& S) {
    if (ASM_LOADS) {
      const char* g0 = gbase + (size_t)((unsigned)r0 * (unsigned)(WIDTH * 4));  // rows r0 + {0, 2}: immediates 0 / 2048 (+ row h in voff)
      const char* g1 = g0 + 4 * (WIDTH * 4);                                     // rows r0 + {4, 6}
      const char* x0 = xbase + (size_t)((unsigned)r0 * (unsigned)(WIDTH * 4));
      const char* x1 = x0 + 4 * (WIDTH * 4);
      S.a[0] = gload_x4<0>(voff_a, g0);
      S.a[1] = gload_x4<2 * WIDTH * 4>(voff_a, g0);
      S.a[2] = gload_x4<0>(voff_a, g1);
      S.a[3] = gload_x4<2 * WIDTH * 4>(voff_a, g1);
      S.b[0] = gload_x2<0>(voff_b, x0);
      S.b[1] = gload_x2<2 * WIDTH * 4>(voff_b, x0);
      S.b[2] = gload_x2<0>(voff_b, x1);
      S.b[3] = gload_x2<2 * WIDTH * 4>(voff_b, x1);
      return;
    }
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) {
      int r = r0 + 2 * u + h;
      if (CHECK) r = r < r_end ? r : r_end - 1;
      const char* ga = gbase + (size_t)((unsigned)r * ga_row_bytes) + (voff_a - (unsigned)h * ga_row_bytes);
      const char* xb = xbase + (size_t)((unsigned)r * (unsigned)(WIDTH * 4)) + (voff_b - (unsigned)h * (WIDTH * 4));
      if (NCA == 4) S.a[u] = *reinterpret_cast<const float4*>(ga);
      else S.a[u].x = *reinterpret_cast<const float*>(ga);
      S.b[u] = *reinterpret_cast<const float2*>(xb);
    }
  };
  // ASM_LOADS: wait until this stage has landed (the two younger stages = 16 loads may still be in flight)
  auto landed = [&](DwStage<NCA>& S) {
    if (ASM_LOADS) {
      f32x4 a[4];
      f32x2 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = f32x4{S.a[u].x, S.a[u].y, S.a[u].z, S.a[u].w};
        b[u] = f32x2{S.b[u].x, S.b[u].y};
      }
      asm volatile("s_waitcnt vmcnt(16)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        S.a[u] = make_float4(a[u][0], a[u][1], a[u][2], a[u][3]);
        S.b[u] = make_float2(b[u][0], b[u][1]);
      }
    }
  };
  auto mul_u = [&](int r0, const DwStage<NCA>& S, int u) {
    float4 a = S.a[u];
    if (NCA == 1) a.x = a_live ? a.x : 0.f;  // thin heads: lanes q >= 4 supply zero rows
    if (CHECK) {
      if (r0 + 2 * u + h >= r_end) a = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the end contribute nothing
    }
#pragma unroll
    for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a, ca), comp(S.b[u], cb), acc[ca][cb], 0, 0, 0);
    if (BIAS) {
      bsum[0] += a.x;
      if (NCA == 4) { bsum[1] += a.y; bsum[2] += a.z; bsum[3] += a.w; }
    }
  };
  auto mul = [&](int r0, const DwStage<NCA>& S) {
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) mul_u(r0, S, u);
  };
  constexpr int G = DW_ROWS;
  DwStage<NCA> s0, s1, s2;
  // branch-free rotation: the two prefetches past the end re-read the last stage (valid memory, never multiplied)
  const int r_last = CHECK ? r_end : r_end - G;
  auto at = [&](int r) { return (CHECK || r <= r_last) ? r : r_last; };
  // ASM_LOADS: the two stages that stay in flight across the loop's back edge must have LANDED there -- the compiler is free to
  // copy a loop-carried register at the edge (it did: v_mov of a stage whose loads were still in flight = stale operands on
  // some rounds), and an asm load's destination counts as written at the statement.  So every round (3 stages, 96 MFMAs)
  // ends with one full wait; inside the round the waits stay counted, two stages ahead.
  auto landed_all = [&](DwStage<NCA>& A, DwStage<NCA>& B) {
    if (ASM_LOADS) {
      f32x4 a[8];
      f32x2 b[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = f32x4{A.a[u].x, A.a[u].y, A.a[u].z, A.a[u].w};
        b[u] = f32x2{A.b[u].x, A.b[u].y};
        a[4 + u] = f32x4{B.a[u].x, B.a[u].y, B.a[u].z, B.a[u].w};
        b[4 + u] = f32x2{B.b[u].x, B.b[u].y};
      }
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]),
                     "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                   :
                   : "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        A.a[u] = make_float4(a[u][0], a[u][1], a[u][2], a[u][3]);
        A.b[u] = make_float2(b[u][0], b[u][1]);
        B.a[u] = make_float4(a[4 + u][0], a[4 + u][1], a[4 + u][2], a[4 + u][3]);
        B.b[u] = make_float2(b[4 + u][0], b[4 + u][1]);
      }
    }
  };
  load(r_begin, s0);
  load(at(r_begin + G), s1);
  landed_all(s0, s1);
  for (int r0 = r_begin; r0 < r_end; r0 += 3 * G) {
    // the scheduling barriers keep each stage's requests where they are written: two stages (64 MFMAs) ahead of
    // their use -- left alone the compiler sinks them next to the uses and every iteration waits on HBM
    load(at(r0 + 2 * G), s2);
    __builtin_amdgcn_sched_barrier(0);
    mul(r0, s0);  // landed at the previous round's end
    __builtin_amdgcn_sched_barrier(0);
    load(at(r0 + 3 * G), s0);
    __builtin_amdgcn_sched_barrier(0);
    mul(r0 + G, s1);  // landed at the previous round's end
    __builtin_amdgcn_sched_barrier(0);
    load(at(r0 + 4 * G), s1);
    __builtin_amdgcn_sched_barrier(0);
    landed(s2);  // younger: the 16 loads of s0, s1
    mul(r0 + 2 * G, s2);
    __builtin_amdgcn_sched_barrier(0);
    landed_all(s0, s1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

// geometry of one product, shared by the kernel, the reduce and the host
__host__ __device__ inline int dwi_in_blocks(const DwItem& p) { return p.thin ? 6 : p.nin / 64; }
__host__ __device__ inline int dwi_nblocks(const DwItem& p) { return p.thin ? 6 : (p.nout / 128) * (p.nin / 64); }  // 8, 4, 2 (thin: 6 of 8 waves)
__host__ __device__ inline int dwi_msubs(const DwItem& p) { return p.thin ? 1 : DW_WAVES / dwi_nblocks(p); }
__host__ __device__ inline size_t dwi_wave_floats(const DwItem& p) { return (size_t)(p.thin ? 1 : 4) * 2 * 16 * 64; }
// per workgroup: 8 wave blocks + 8 x 128 column sums
__host__ __device__ inline size_t dwi_wg_floats(const DwItem& p) { return DW_WAVES * (dwi_wave_floats(p) + 128); }

template <int NCA>
__global__ __launch_bounds__(512, 2) void k_dw(const DwItem p, const long long Mtot, float* __restrict__ slabs) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform values stay in SGPRs from here on
#ifdef NERF_STAMPS
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
  const int in_blocks = dwi_in_blocks(p);
  const int nblocks = dwi_nblocks(p);
  const int msubs = dwi_msubs(p);
  const int blk = wv % nblocks, msub = wv / nblocks;
  if (NCA == 1 && wv >= nblocks) return;  // thin heads: waves 6, 7 have no block
  const int oA = NCA == 1 ? 0 : (blk / in_blocks) * 128;
  const int gran = DW_ROWS * DW_STAGES * msubs;
  const int Mrows = (int)Mtot;  // < 2^31 rows and < 4 GiB per operand (checked by the host)
  const int per_wg = ((Mrows + DW_WGS - 1) / DW_WGS + gran - 1) / gran * gran;
  const int per_wave = per_wg / msubs;  // a multiple of the 24 rows of one pipeline round
  const long long r_begin64 = (long long)blockIdx.x * per_wg + (long long)msub * per_wave;
  const int r_begin = r_begin64 < Mrows ? (int)r_begin64 : Mrows;
  const long long r_nom = r_begin64 + per_wave;
  const int r_end = r_nom > Mrows ? Mrows : (int)r_nom;
  const int h = lane >> 5, q = lane & 31;

  f32x16 acc[NCA][2];
#pragma unroll
  for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // operand bases (uniform) and the one per-lane offset of each operand
  const char* gbase;
  const char* xbase;
  unsigned ga_row_bytes, voff_a;
  bool do_bias;  // wave-uniform: column sums of A on this wave
  if (NCA == 4) {
    gbase = reinterpret_cast<const char*>(p.G + oA);
    xbase = reinterpret_cast<const char*>(p.X + (blk % in_blocks) * 64);
    ga_row_bytes = WIDTH * 4;
    voff_a = (unsigned)h * (WIDTH * 4) + (unsigned)q * 16;
    // column sums of G = bias gradient: one of the in_blocks waves that read the same G columns does them (4 adds per k-step;
    // two such waves per workgroup, on different SIMDs)
    do_bias = p.db != nullptr && (blk % in_blocks) == 0;
  } else {
    gbase = reinterpret_cast<const char*>(p.G);  // [rows][4]
    xbase = reinterpret_cast<const char*>(blk < 4 ? p.X + blk * 64 : p.X2 + (blk - 4) * 64);
    ga_row_bytes = 16;
    voff_a = (unsigned)h * 16 + (unsigned)(q & 3) * 4;
    do_bias = blk == 0;
  }
  const unsigned voff_b = (unsigned)h * (WIDTH * 4) + (unsigned)q * 8;
  const bool a_live = q < 4;
  if (r_begin < r_end) {
    const bool full = r_nom <= Mrows;
    if (full && do_bias) dw_rows<NCA, false, true>(gbase, xbase, ga_row_bytes, voff_a, voff_b, r_begin, r_end, h, a_live, acc, bsum);
    else if (full) dw_rows<NCA, false, false>(gbase, xbase, ga_row_bytes, voff_a, voff_b, r_begin, r_end, h, a_live, acc, bsum);
    else if (do_bias) dw_rows<NCA, true, true>(gbase, xbase, ga_row_bytes, voff_a, voff_b, r_begin, r_end, h, a_live, acc, bsum);
    else dw_rows<NCA, true, false>(gbase, xbase, ga_row_bytes, voff_a, voff_b, r_begin, r_end, h, a_live, acc, bsum);
  }
  // this wave's slab block and its 128 column sums (zeros where it did not sum)
  float* wg = slabs + p.slab_off + (size_t)blockIdx.x * dwi_wg_floats(p);
  float* ws = wg + (size_t)wv * dwi_wave_floats(p);
#pragma unroll
  for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[((size_t)(ca * 2 + cb) * 16 + r) * 64 + lane] = acc[ca][cb][r];
#pragma unroll
  for (int c = 0; c < 4; ++c) bsum[c] += __shfl_xor(bsum[c], 32);
  if (h == 0) {
    float* bs = wg + DW_WAVES * dwi_wave_floats(p) + (size_t)wv * 128;
    if (NCA == 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[4 * q + c] = bsum[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[4 * q + c] = (c == 0 && q < 4) ? bsum[0] : 0.f;  // entry 4q: column sum of A column q
    }
  }
#ifdef NERF_STAMPS
  if (p.stamps && lane == 0) {  // per wave: 100 MHz timestamps, XCC and hardware ids (workgroup = blockIdx.x, wave wv)
    unsigned long long* r = p.stamps + ((size_t)blockIdx.x * 8 + wv) * 4;
    r[0] = t_start;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    r[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  }
#endif
}

// Sums the slabs of every product of the step and scatters into the nn.Linear-layout gradients: grid (blocks, items).
__global__ __launch_bounds__(256) void k_dw_reduce(const DwBatch b) {
  const DwItem& p = b.item[blockIdx.y];
  const int nblocks = dwi_nblocks(p), msubs = dwi_msubs(p), in_blocks = dwi_in_blocks(p);
  const size_t wave_floats = dwi_wave_floats(p), wg_floats = dwi_wg_floats(p);
  const int n_w = nblocks * (int)wave_floats;  // weight elements (padded)
  const int n_b = p.thin ? 4 : (p.db ? p.nout : 0);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_w + n_b) return;
  const float* base = b.slabs + p.slab_off;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n_w) {
    const int blk = e / (int)wave_floats;
    int r = e - blk * (int)wave_floats;
    for (int ms = 0; ms < msubs; ++ms) {  // fixed order: msub, then workgroup, four partial sums
      const float* q = base + (size_t)(ms * nblocks + blk) * wave_floats + r;
      for (int k = 0; k < DW_WGS; k += 4) {
        s0 += q[(size_t)k * wg_floats]; s1 += q[(size_t)(k + 1) * wg_floats];
        s2 += q[(size_t)(k + 2) * wg_floats]; s3 += q[(size_t)(k + 3) * wg_floats];
      }
    }
    const float s = (s0 + s1) + (s2 + s3);
    const int lane = r & 63; r >>= 6;
    const int reg = r & 15; r >>= 4;
    const int cb = r & 1, ca = r >> 1;
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);  // row of the 32 x 32 tile
    if (p.thin) {
      const int out = i;                                  // row of A: 0..2 colour, 3 sigma
      const int in = blk * 64 + 2 * (lane & 31) + cb;     // 0..255: h7 column, 256..383: c column
      if (in < WIDTH) { if (out == 3) p.dW2[in] = s; }    // dw_sigma[256]
      else if (out < 3) p.dW[(size_t)out * HALF + (in - WIDTH)] = s;  // dW_color[3][128]
    } else {
      const int oA = (blk / in_blocks) * 128, iB = (blk % in_blocks) * 64;
      const int out = oA + 4 * i + ca;
      const int in = iB + 2 * (lane & 31) + cb;
      if (in < p.nin_real) p.dW[(size_t)out * p.ldw + p.col0 + in] = s;
    }
  } else {
    const int o = e - n_w;  // column of G (thin: column of A)
    const float* bs = base + DW_WAVES * wave_floats;
    if (p.thin) {
      for (int k = 0; k < DW_WGS; k += 2) { s0 += bs[(size_t)k * wg_floats + 4 * o]; s1 += bs[(size_t)(k + 1) * wg_floats + 4 * o]; }
      const float s = s0 + s1;
      if (o < 3) p.db[o] = s; else p.db2[0] = s;  // db_color[3], db_sigma
    } else {
      const int ob = o / 128, oi = o % 128;
      for (int w = 0; w < DW_WAVES; ++w) {
        if (((w % nblocks) / in_blocks) != ob || ((w % nblocks) % in_blocks) != 0) continue;  // the waves that summed these G columns
        const float* q = bs + (size_t)w * 128 + oi;
        for (int k = 0; k < DW_WGS; k += 4) {
          s0 += q[(size_t)k * wg_floats]; s1 += q[(size_t)(k + 1) * wg_floats];
          s2 += q[(size_t)(k + 2) * wg_floats]; s3 += q[(size_t)(k + 3) * wg_floats];
        }
      }
      p.db[o] = (s0 + s1) + (s2 + s3);
    }
  }
}

size_t dw_item_slab_floats(const DwItem& p) { return (size_t)DW_WGS * dwi_wg_floats(p); }

hipError_t launch_dw(const DwItem& p, long long Mtot, float* slabs, hipStream_t st) {
  if (p.thin)
    hipLaunchKernelGGL((k_dw<1>), dim3(DW_WGS), dim3(512), 0, st, p, Mtot, slabs);
  else
    hipLaunchKernelGGL((k_dw<4>), dim3(DW_WGS), dim3(512), 0, st, p, Mtot, slabs);
  return hipGetLastError();
}

hipError_t launch_dw_reduce(const DwBatch& b, hipStream_t st) {
  int most = 0;
  for (int i = 0; i < b.n; ++i) {
    const DwItem& p = b.item[i];
    const int total = dwi_nblocks(p) * (int)dwi_wave_floats(p) + (p.thin ? 4 : (p.db ? p.nout : 0));
    most = total > most ? total : most;
  }
  hipLaunchKernelGGL(k_dw_reduce, dim3((most + 255) / 256, b.n), dim3(256), 0, st, b);
  return hipGetLastError();
}

}  // namespace nerf
