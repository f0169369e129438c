// dw_f32.hip -- the weight-gradient GEMMs of the fp32 train step for MI355X (gfx950).
//
//  k_dw4<NCB>     one product: dW[out][in] = sum_m G[m][out] * X[m][in] as a split-M fp32 MFMA GEMM
//                 (v_mfma_f32_32x32x2_f32 with the SAMPLE as the k index): both operands are read straight from their
//                 row-major HBM images (lane (q, h) <- columns 4q..4q+3 of G and NCB q.. of X, row m + h).  ONE wave per SIMD:
//                 a 128 x 32 NCB output block = 64 NCB accumulators per wave, 4 waves = one workgroup per CU on 1/256 of the
//                 rows, an 8-deep register ring of k-steps (7 requested ahead), per-wave partial slabs.  fp32 MFMA runs on the
//                 SIMD's fp32 lanes, so every VALU instruction in the loop is MFMA time lost: a wave carries at most one extra
//                 duty -- the column sums of its G block (bias gradient) or, on the point_info product, the sigma head's weight
//                 gradient (sum_m dsigma_pre[m] * h7[m][:], same X) -- as asm-pinned adds / FMAs in the same loop.
//  k_dw4_group    the seven 256 x 256 products (layers 1..7) in ONE launch, 36 workgroups each: same body, a seventh of the slabs
//  k_dw_thin      the colour head as one thin product: A = the [rows][4] buffer (dz_r, dz_g, dz_b, dsigma_pre), one 32-row
//                 output tile per wave, X = c; the column sums of A are the bias gradients of both heads.  HBM-bound.
//  k_dw_reduce    ONE launch per step: sums the slabs of all products in a fixed order (deterministic, no float atomics)
//                 and scatters into the nn.Linear-layout gradients.
//  k_fold_grads   point_info is folded into dir_info (common.h SEG_FOLD): ONE 128 x 256 product (dpre_dir^T h7) replaces the
//                 256 x 256 point_info product and dir_info's feature columns; this kernel turns it into the two tensors' gradients.
//
// Autograd spans replaced: the weight / bias gradients of Network.forward (nerf.py:101-124) as produced by
// loss.backward() at nerf.py:473.
#include "field_common.h"
#include "bf16_stream.h"  // static_for

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

// Main loop of the thin product (the 3-stage rotation of round 1's two-waves-per-SIMD kernel; NCA = 1 is the only instantiation
// left) over rows [r_begin, r_end) (wave-uniform).  CHECK = false: every row of every stage is in range (no selects
// between the loads and their use, so the loads of two stages stay in flight behind the MFMAs of the third).  CHECK = true
// (the ragged tail of a pass only): rows are clamped per lane and rows past the end contribute nothing.
// gp / xp: this lane's operand pointers at row 0 (column group q; row h is added per stage).
// (Scalar-base loads -- `global_load v, v_off32, s[base:base+1]`, no vector address arithmetic at all -- were built as inline
// asm and measured 2 % faster, but an asm load's destination is invisible to the register allocator: it placed copies of
// in-flight stage registers in front of the waits on some code paths, i.e. stale operands.  Plain loads it is.)
template <int NCA, bool CHECK>
__device__ __forceinline__ void dw_rows(const bool BARRIER, const bool BIAS, const float* __restrict__ gp, const float* __restrict__ xp, int ga_row_floats, long long r_begin,
                                        long long r_end, int h, bool a_live, f32x16 (&acc)[NCA][2], float (&bsum)[4]) {
  auto load = [&](long long r0, DwStage<NCA>& S) {
    if (!CHECK) {
      // one 64-bit offset per stage and operand; the four row pairs are immediates of it
      const float* ga = gp + (size_t)(r0 + h) * ga_row_floats;
      const float* xb = xp + (size_t)(r0 + h) * WIDTH;
#pragma unroll
      for (int u = 0; u < DW_UNROLL; ++u) {
        if (NCA == 4) S.a[u] = *reinterpret_cast<const float4*>(ga + (size_t)u * 2 * WIDTH);
        else S.a[u].x = ga[u * 2 * 4];
        S.b[u] = *reinterpret_cast<const float2*>(xb + (size_t)u * 2 * WIDTH);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) {
      long long r = r0 + 2 * u + h;
      r = r < r_end ? r : r_end - 1;
      if (NCA == 4) S.a[u] = *reinterpret_cast<const float4*>(gp + (size_t)r * ga_row_floats);
      else S.a[u].x = gp[(size_t)r * ga_row_floats];
      S.b[u] = *reinterpret_cast<const float2*>(xp + (size_t)r * WIDTH);
    }
  };
  auto mul_u = [&](long long r0, const DwStage<NCA>& S, int u) {
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
  auto mul = [&](long long r0, const DwStage<NCA>& S) {
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) mul_u(r0, S, u);
  };
  constexpr long long G = DW_ROWS;
  if (CHECK) {  // ragged tail of a pass: one stage at a time, no pipelining (a few waves of one launch take this path)
    DwStage<NCA> s;
    for (long long r0 = r_begin; r0 < r_end; r0 += G) {
      load(r0, s);
      mul(r0, s);
    }
    return;
  }
  DwStage<NCA> s0, s1, s2;
  // branch-free rotation: the two prefetches past the end re-read the last stage (valid memory, never multiplied)
  const long long r_last = r_end - G;
  auto at = [&](long long r) { return r <= r_last ? r : r_last; };
  load(r_begin, s0);
  load(at(r_begin + G), s1);
  unsigned round = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += 3 * G) {
    // Keep the workgroup's waves on the same rows: the 4 (or 2) waves that share an operand block read it from L1/L2 only
    // while they stay within a few rounds of each other.  Unfenced, the older wave of every SIMD pair wins each MFMA
    // arbitration and runs ahead (measured: it finishes at 78 % of the kernel's time); over a 28,000-row range that drift
    // is megabytes, every wave then streams its operands from HBM on its own, and the phase becomes HBM-bound (3x slower).
    // Every wave of a workgroup that gets here runs the same number of rounds (BARRIER is workgroup-uniform).
    if (BARRIER && (round++ & 1) == 0) __builtin_amdgcn_s_barrier();
    // the scheduling barriers keep each stage's requests where they are written: two stages (64 MFMAs) ahead of
    // their use -- left alone the compiler sinks them next to the uses and every iteration waits on HBM
    load(at(r0 + 2 * G), s2);
    __builtin_amdgcn_sched_barrier(0);
    mul(r0, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(at(r0 + 3 * G), s0);
    __builtin_amdgcn_sched_barrier(0);
    mul(r0 + G, s1);
    __builtin_amdgcn_sched_barrier(0);
    load(at(r0 + 4 * G), s1);
    __builtin_amdgcn_sched_barrier(0);
    mul(r0 + 2 * G, s2);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- one wave per SIMD: a 128 x (32 NCB) output block per wave (256 / 128 accumulators), no SIMD partner ---------------------
// Two waves per SIMD do not share the matrix pipe evenly (the older wave wins every arbitration, ends at 78 % of the kernel's
// time and leaves its partner latency-bound).  Here a SIMD has ONE wave, which hides the memory
// latency itself: a ring of DW4_DEPTH k-steps (one 16-byte load per operand and k-step), DW4_DEPTH - 1 of them (7 x 1024 MFMA
// cycles = 3 us) requested ahead of the one being multiplied.  Per k-step: 2 vector-memory instructions for 16 (8) MFMAs.
#ifndef DW4_DEPTH_V
#define DW4_DEPTH_V 8
#endif
constexpr int DW4_DEPTH = DW4_DEPTH_V;

template <int NCB> struct DwVecB;
template <> struct DwVecB<4> { typedef float4 type; };
template <> struct DwVecB<2> { typedef float2 type; };

template <int NCB>
struct DwFrag {
  float4 a;
  typename DwVecB<NCB>::type b;
  float sg;  // DW_SIG only: dsigma_pre of the lane's row
};

// what a wave does besides its block (one extra duty per wave at most, so that the four waves of a workgroup stay level):
constexpr int DW_PLAIN = 0, DW_BIAS = 1, DW_SIG = 2, DW_BIAS_LO = 3, DW_BIAS_HI = 4, DW_RAY_LO = 5, DW_RAY_HI = 6;
// the 128 x 256 product of dpre_dir with h7 (point_info folded into dir_info, common.h SEG_FOLD) has only two waves per operand block
// and three duties -- the sigma head (X = h7), dir_info's bias gradient and the per-ray sums: every wave carries the sigma head for ITS
// 128 columns of h7 and half of the column sums (6 VALU + one 4-byte load per 16 MFMAs: 2.6 % of ONE product)
constexpr int DW_SIG_BIAS_LO = 7, DW_SIG_BIAS_HI = 8, DW_SIG_RAY_LO = 9, DW_SIG_RAY_HI = 10;
__host__ __device__ constexpr bool duty_sig(int d) { return d == DW_SIG || d >= DW_SIG_BIAS_LO; }
__host__ __device__ constexpr bool duty_ray(int d) { return d == DW_RAY_LO || d == DW_RAY_HI || d == DW_SIG_RAY_LO || d == DW_SIG_RAY_HI; }
__host__ __device__ constexpr bool duty_lo(int d) { return d == DW_BIAS_LO || d == DW_RAY_LO || d == DW_SIG_BIAS_LO || d == DW_SIG_RAY_LO; }
__host__ __device__ constexpr bool duty_hi(int d) { return d == DW_BIAS_HI || d == DW_RAY_HI || d == DW_SIG_BIAS_HI || d == DW_SIG_RAY_HI; }
// DW_BIAS  column sums of its 128 columns of G = bias gradient (4 adds per k-step)
// DW_BIAS_LO / _HI  the same sums shared by the two waves that read the same G block: columns 4q, 4q+1 / 4q+2, 4q+3 (2 adds per k-step)
// DW_RAY_LO / _HI   as DW_BIAS_LO / _HI, and the sums of every ray's rows are written out on the way (dir_info: the gamma_d columns need them)
struct RayDuty {
  float* out;        // [2][rays][128]
  int nc, nf, rows_c;
  int rays;
};
// DW_SIG   sigma head: wsig[col] += dsigma_pre[row] * X[row][col] for its 32 NCB columns of X (one more 4-byte load and 4 FMAs per k-step)

// rows [r_begin, r_end) (wave-uniform, a multiple of 2 DW4_DEPTH rows long).  gbase / xbase / sbase: wave-uniform operand
// pointers at row 0; goff / xoff / soff: this lane's byte offset inside a row pair (row h, column group q).
// bsum: the duty's sums (column sums of G; DW_SIG alone: the sigma head's); ssum: the sigma head's sums of the combined duties
template <int NCB, int DUTY>
__device__ __forceinline__ void dw_stream(const float* __restrict__ gbase, const float* __restrict__ xbase, const float* __restrict__ sbase, const unsigned goff,
                                          const unsigned xoff, const unsigned soff, const int r_begin, const int r_end, f32x16 (&acc)[4][NCB], float (&bsum)[4],
                                          const RayDuty rd = RayDuty{nullptr, 0, 0, 0, 0}, const int lane = 0, float* ssum = nullptr) {
  typedef typename DwVecB<NCB>::type VB;
  constexpr int D = DW4_DEPTH;
  constexpr bool RAY = duty_ray(DUTY);
  constexpr bool SIG2 = DUTY >= DW_SIG_BIAS_LO;  // sigma head next to column sums: its sums live in ssum
  DwFrag<NCB> s[D];
  float rs[2] = {0.f, 0.f};  // RAY: this lane's sums over the rows of the current ray (its two columns, its row parity)
  const int r_last = r_end - 2;
  auto load = [&](int r, DwFrag<NCB>& S) {
    r = r <= r_last ? r : r_last;  // the prefetches past the end re-read the last row pair (valid memory, never multiplied)
    const char* ga = reinterpret_cast<const char*>(gbase + (size_t)r * WIDTH);
    const char* xa = reinterpret_cast<const char*>(xbase + (size_t)r * WIDTH);
    unsigned go = goff, xo = xoff;
    asm volatile("" : "+v"(go), "+v"(xo));  // opaque here: otherwise base + lane offset is hoisted as a 64-bit vector and every load pays a 64-bit vector add
    S.a = *reinterpret_cast<const float4*>(ga + go);
    S.b = *reinterpret_cast<const VB*>(xa + xo);
    if (duty_sig(DUTY)) S.sg = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase + (size_t)r * 4) + soff);
  };
  auto mul = [&](const DwFrag<NCB>& S) {
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(S.a, ca), comp(S.b, cb), acc[ca][cb], 0, 0, 0);
    // the duties as asm statements: plain adds get re-associated across the ring, the stage loads follow them, and the loop
    // ends up requesting all eight stages at once
    if (DUTY == DW_BIAS) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[0]) : "v"(S.a.x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[1]) : "v"(S.a.y));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[2]) : "v"(S.a.z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[3]) : "v"(S.a.w));
    }
    if (DUTY == DW_SIG) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(bsum[cb]) : "v"(S.sg), "v"(comp(S.b, cb)));
    }
    if (DUTY == DW_BIAS_LO) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[0]) : "v"(S.a.x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[1]) : "v"(S.a.y));
    }
    if (DUTY == DW_BIAS_HI) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[2]) : "v"(S.a.z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[3]) : "v"(S.a.w));
    }
    if (DUTY == DW_RAY_LO || DUTY == DW_SIG_RAY_LO) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(rs[0]) : "v"(S.a.x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(rs[1]) : "v"(S.a.y));
    }
    if (DUTY == DW_RAY_HI || DUTY == DW_SIG_RAY_HI) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(rs[0]) : "v"(S.a.z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(rs[1]) : "v"(S.a.w));
    }
    if (DUTY == DW_SIG_BIAS_LO) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[0]) : "v"(S.a.x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[1]) : "v"(S.a.y));
    }
    if (DUTY == DW_SIG_BIAS_HI) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[2]) : "v"(S.a.z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[3]) : "v"(S.a.w));
    }
    if (SIG2) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ssum[cb]) : "v"(S.sg), "v"(comp(S.b, cb)));
    }
  };
  // RAY: which ray the range starts in and how many of its rows are left (ranges start and end on ray boundaries: dw_ray_duty_ok)
  bool fine = RAY && r_begin >= rd.rows_c;
  int ray = !RAY ? 0 : (fine ? (r_begin - rd.rows_c) / rd.nf : r_begin / rd.nc);
  int left = fine ? rd.nf : rd.nc;
  static_for<D - 1>([&](auto I) {  // in ring order (left alone the scheduler issues them last to first, and the loop's first wait drains the ring)
    load(r_begin + 2 * (int)I, s[I]);
    __builtin_amdgcn_sched_barrier(0);
  });
  for (int r0 = r_begin; r0 < r_end; r0 += 2 * D) {
    static_for<D>([&](auto I) {
      constexpr int i = I;
      load(r0 + 2 * (i + D - 1), s[(i + D - 1) % D]);
      __builtin_amdgcn_sched_barrier(0);
      mul(s[i]);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (RAY) {  // (wave-uniform) a ray's rows end with this round: its sums go out, the column sums take them over
      left -= 2 * D;
      if (left == 0) {
        constexpr int c0 = duty_lo(DUTY) ? 0 : 2;
        bsum[c0] += rs[0];
        bsum[c0 + 1] += rs[1];
        const float t0 = rs[0] + __shfl_xor(rs[0], 32), t1 = rs[1] + __shfl_xor(rs[1], 32);
        if (lane < 32) *reinterpret_cast<float2*>(rd.out + ((size_t)(fine ? rd.rays : 0) + ray) * 128 + 4 * lane + c0) = make_float2(t0, t1);
        rs[0] = 0.f;
        rs[1] = 0.f;
        ++ray;
        if (!fine && r0 + 2 * D == rd.rows_c) {  // the coarse pass's rows end here; the fine pass's follow
          fine = true;
          ray = 0;
        }
        left = fine ? rd.nf : rd.nc;
      }
    }
  }
}

// ragged tail of a pass (a few waves of a launch): one row pair at a time, rows clamped per lane, rows past the end contribute 0
template <int NCB>
__device__ __forceinline__ void dw_stream_tail(const float* __restrict__ gbase, const float* __restrict__ xbase, const float* __restrict__ sbase, const unsigned goff,
                                               const unsigned xoff, const unsigned soff, const int h, const int r_begin, const int r_end, const int duty,
                                               f32x16 (&acc)[4][NCB], float (&bsum)[4], float* ssum = nullptr) {
  typedef typename DwVecB<NCB>::type VB;
  for (int r0 = r_begin; r0 < r_end; r0 += 2) {
    const bool live = r0 + h < r_end;
    const int r = live ? r0 : r_end - 1 - h;  // (row r + h of the lane's offsets = r_end - 1)
    float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gbase + (size_t)r * WIDTH) + goff);
    const VB b = *reinterpret_cast<const VB*>(reinterpret_cast<const char*>(xbase + (size_t)r * WIDTH) + xoff);
    float sg = 0.f;
    if (duty_sig(duty) && live) sg = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase + (size_t)r * 4) + soff);
    if (!live) a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a, ca), comp(b, cb), acc[ca][cb], 0, 0, 0);
    if (duty == DW_BIAS) { bsum[0] += a.x; bsum[1] += a.y; bsum[2] += a.z; bsum[3] += a.w; }
    if (duty_lo(duty)) { bsum[0] += a.x; bsum[1] += a.y; }  // (the ray duties never get here: dw_ray_duty_ok)
    if (duty_hi(duty)) { bsum[2] += a.z; bsum[3] += a.w; }
    if (duty == DW_SIG) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bsum[cb] = __builtin_fmaf(sg, comp(b, cb), bsum[cb]);
    }
    if (duty >= DW_SIG_BIAS_LO) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) ssum[cb] = __builtin_fmaf(sg, comp(b, cb), ssum[cb]);
    }
  }
}

}  // namespace

// geometry of one product, shared by the kernels, the reduce and the host
__host__ __device__ inline int dwi_ncb(const DwItem& p) { return p.thin ? 2 : (p.nin % 128 == 0 ? 4 : 2); }  // 32-column tiles of X per wave
__host__ __device__ inline int dwi_waves(const DwItem& p) { return p.thin ? DW_WAVES : 4; }
__host__ __device__ inline int dwi_in_blocks(const DwItem& p) { return p.nin / (32 * dwi_ncb(p)); }
__host__ __device__ inline int dwi_nblocks(const DwItem& p) { return p.thin ? dwi_in_blocks(p) : (p.nout / 128) * dwi_in_blocks(p); }  // 4, 2 (thin: 2)
__host__ __device__ inline int dwi_msubs(const DwItem& p) { return dwi_waves(p) / dwi_nblocks(p); }
__host__ __device__ inline size_t dwi_wave_floats(const DwItem& p) { return (size_t)(p.thin ? 1 : 4) * dwi_ncb(p) * 16 * 64; }
// per workgroup: one block per wave, 128 column sums per wave, and (has_sig) 128 sigma-head sums per wave
__host__ __device__ inline size_t dwi_wg_floats(const DwItem& p) { return dwi_waves(p) * (dwi_wave_floats(p) + 128 + (p.has_sig ? 128 : 0)); }
// Slab layout of one product (floats from slab_off): first the accumulator rows -- row = 64 consecutive floats of one wave's
// block (wave, tile, register) -- stored [workgroup][row][64]: every workgroup writes one contiguous run (the transposed order,
// [row][workgroup][64], lets the reduce read contiguously but costs the products more than it saves: their 256 stores per wave
// then lie 64 KiB apart) -- then per workgroup the waves' column sums ([wave][128], and [wave][128] more for the sigma head).
__host__ __device__ inline size_t dwi_rows_per_wg(const DwItem& p) { return dwi_waves(p) * dwi_wave_floats(p) / 64; }
__host__ __device__ inline size_t dwi_row_off(const DwItem& p, size_t row, int lw) { return ((size_t)lw * dwi_rows_per_wg(p) + row) * 64; }
__host__ __device__ inline size_t dwi_wg_stride(const DwItem& p) { return dwi_rows_per_wg(p) * 64; }  // floats between two workgroups' copies of a row
__host__ __device__ inline size_t dwi_sums_per_wg(const DwItem& p) { return (size_t)dwi_waves(p) * (128 + (p.has_sig ? 128 : 0)); }
__host__ __device__ inline size_t dwi_sums_off(const DwItem& p, int lw) { return (size_t)p.nwg * dwi_waves(p) * dwi_wave_floats(p) + (size_t)lw * dwi_sums_per_wg(p); }
// the extra duty of the wave that holds block (bi, bj) = (128 columns of G, 32 NCB columns of X); thin: the waves of X block 0 sum A
__host__ __device__ inline int dwi_duty(const DwItem& p, int bi, int bj) {
  const int in_blocks = dwi_in_blocks(p);
  if (p.thin) return bj == 0 ? 1 : 0;
  if (p.has_sig && p.nout == 128) {  // dpre_dir^T h7: sigma head on every wave (its own 128 columns of h7) + half of the column sums
    if (p.raysum) return bj == 0 ? 9 : 10;           // DW_SIG_RAY_LO / _HI
    return bj == 0 ? 7 : 8;                          // DW_SIG_BIAS_LO / _HI
  }
  if (p.has_sig) {  // a 256 x 256 product carrying the sigma head: on the diagonal waves, column sums on the other two
    if (bj == bi) return 2;                          // DW_SIG
    return p.db ? 1 : 0;                             // DW_BIAS
  }
  if (!p.db) return 0;
  if (in_blocks == 2 && p.raysum) return bj == 0 ? 5 : 6;  // DW_RAY_LO / DW_RAY_HI: the same halves, per-ray sums written on the way
  if (in_blocks == 2) return bj == 0 ? 3 : 4;        // DW_BIAS_LO / DW_BIAS_HI: half the sums on each of the two waves of a G block
  return bj == 0 ? 1 : 0;                            // DW_BIAS (one wave per G block)
}
__host__ __device__ inline bool dwi_sums_columns(int duty) { return duty == 1 || duty >= 3; }
__host__ __device__ inline bool dwi_sums_sigma(int duty) { return duty == 2 || duty >= 7; }

// The colour head as one thin product: A = the [rows][4] buffer (dz_r, dz_g, dz_b, dsigma_pre), one 32-row output tile per wave,
// X = c in two 64-column blocks x four row sub-ranges; rows 0..2 of the result are dW_color, the column sums of A (waves of
// block 0) the bias gradients of both heads.  (The sigma head's weights ride on the point_info product, which reads h7 anyway.)
__global__ __launch_bounds__(512, 2) void k_dw_thin(const DwItem p, const long long Mtot, float* __restrict__ slabs) {
  const int tid = threadIdx.x, lane = tid & 63, lw = blockIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform values stay in SGPRs from here on
  constexpr int nblocks = 2, msubs = DW_WAVES / nblocks;
  const int blk = wv % nblocks, msub = wv / nblocks;
  const int gran = DW_ROWS * DW_STAGES * msubs;
  const int Mrows = (int)Mtot;  // < 2^31 rows (api.hip: check_sizes); operand addresses are 64-bit
  const int per_wg = ((Mrows + p.nwg - 1) / p.nwg + gran - 1) / gran * gran;
  const int per_wave = per_wg / msubs;  // a multiple of the 24 rows of one pipeline round
  const long long r_begin64 = (long long)lw * per_wg + (long long)msub * per_wave;
  const int r_begin = r_begin64 < Mrows ? (int)r_begin64 : Mrows;
  const long long r_nom = r_begin64 + per_wave;
  const int r_end = r_nom > Mrows ? Mrows : (int)r_nom;
  const int h = lane >> 5, q = lane & 31;

  f32x16 acc[1][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][cb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  const float* gp = p.G + (q & 3);  // [rows][4]
  const float* xp = p.X + blk * 64 + 2 * q;
  const bool do_bias = blk == 0;  // wave-uniform: column sums of A on this wave
  const bool a_live = q < 4;
  if (r_begin < r_end) {
    if (r_nom <= Mrows) dw_rows<1, false>(false, do_bias, gp, xp, 4, r_begin, r_end, h, a_live, acc, bsum);
    else dw_rows<1, true>(false, do_bias, gp, xp, 4, r_begin, r_end, h, a_live, acc, bsum);
  }
  // this wave's slab rows and its column sums (entry 4q: column sum of A column q; zeros where it did not sum)
  float* sl = slabs + p.slab_off;
  const size_t row0 = (size_t)wv * (dwi_wave_floats(p) / 64);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) sl[dwi_row_off(p, row0 + cb * 16 + r, lw) + lane] = acc[0][cb][r];
  bsum[0] += __shfl_xor(bsum[0], 32);
  if (h == 0) {
    float* bs = sl + dwi_sums_off(p, lw) + (size_t)wv * 128;
#pragma unroll
    for (int c = 0; c < 4; ++c) bs[4 * q + c] = (c == 0 && q < 4) ? bsum[0] : 0.f;
  }
}

// One workgroup's share of one big product, 4 waves: wave = (block, row sub-range); block = 128 columns of G x 32 NCB columns of X.
// MIXED: the 128 x 256 product that carries the sigma head AND column sums on every wave (DW_SIG_*): a kernel of its own, so that the
// loops of the other products keep the schedule they were tuned with (this compiler schedules the stage loads of a loop differently
// when other loops share its kernel).
template <int NCB, bool MIXED>
__device__ __forceinline__ void dw4_body(const DwItem& p, const long long Mtot, float* __restrict__ slabs, const int lw) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NERF_STAMPS
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
  const int in_blocks = p.nin / (32 * NCB);
  const int nblocks = (p.nout / 128) * in_blocks;
  const int msubs = 4 / nblocks;
  const int blk = wv % nblocks, msub = wv / nblocks;
  const int gran = 2 * DW4_DEPTH * msubs;
  const int Mrows = (int)Mtot;  // < 2^31 rows (api.hip: check_sizes); operand addresses are 64-bit
  const int per_wg = ((Mrows + p.nwg - 1) / p.nwg + gran - 1) / gran * gran;
  const int per_wave = per_wg / msubs;  // a multiple of the 2 DW4_DEPTH rows of one ring round
  const long long r_begin64 = (long long)lw * per_wg + (long long)msub * per_wave;
  const int r_begin = r_begin64 < Mrows ? (int)r_begin64 : Mrows;
  const long long r_nom = r_begin64 + per_wave;
  const int r_end = r_nom > Mrows ? Mrows : (int)r_nom;
  const int h = lane >> 5, q = lane & 31;

  f32x16 acc[4][NCB];
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum[4] = {0.f, 0.f, 0.f, 0.f};  // MIXED: the sigma head's sums (bsum: column sums)

  const int bi = blk / in_blocks, bj = blk % in_blocks;
  const float* gbase = p.G + bi * 128;
  const float* xbase = p.X + bj * (32 * NCB);
  const float* sbase = p.sig;
  const unsigned goff = (unsigned)(h * WIDTH + 4 * q) * 4u, xoff = (unsigned)(h * WIDTH + NCB * q) * 4u, soff = (unsigned)h * 16u;
  const int duty = dwi_duty(p, bi, bj);  // wave-uniform
  // a ragged range (the last workgroups of a launch): whole ring rounds through the pipelined loop, only the rest row pair by row pair
  const int r_full = r_begin + (r_end - r_begin) / (2 * DW4_DEPTH) * (2 * DW4_DEPTH), r_stop = r_end;
  if constexpr (MIXED) {
    if (r_begin < r_end) {
      if (r_begin < r_full) {
        const int r_end = r_full;
        const RayDuty rd{p.raysum, p.ray_nc, p.ray_nf, p.rows_c, p.rows_c / (p.ray_nc > 0 ? p.ray_nc : 1)};
        if (duty == DW_SIG_RAY_LO) {
          asm volatile("; sigma head + per-ray sums, low half" ::: "memory");
          dw_stream<NCB, MIXED ? DW_SIG_RAY_LO : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane, ssum);
        } else if (duty == DW_SIG_RAY_HI) {
          asm volatile("; sigma head + per-ray sums, high half" ::: "memory");
          dw_stream<NCB, MIXED ? DW_SIG_RAY_HI : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane, ssum);
        } else if (duty == DW_SIG_BIAS_LO) {
          asm volatile("; sigma head + column sums, low half" ::: "memory");
          dw_stream<NCB, MIXED ? DW_SIG_BIAS_LO : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane, ssum);
        } else {
          asm volatile("; sigma head + column sums, high half" ::: "memory");
          dw_stream<NCB, MIXED ? DW_SIG_BIAS_HI : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane, ssum);
        }
      }
      if (r_full < r_stop) dw_stream_tail<NCB>(gbase, xbase, sbase, goff, xoff, soff, h, r_full, r_stop, duty, acc, bsum, ssum);
    }
  } else
  if (r_begin < r_end) {
    if (r_begin < r_full) {
      const int r_end = r_full;
      // (the empty asm statements differ on purpose: identical starts of the branches get hoisted into this block, and the
      // loops then wait for the whole ring at every k-step)
      if (duty == DW_BIAS) {
        asm volatile("; column sums" ::: "memory");
        dw_stream<NCB, DW_BIAS>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum);
      } else if (NCB == 4 && duty == DW_BIAS_LO) {
        asm volatile("; column sums, low half" ::: "memory");
        dw_stream<NCB, NCB == 4 ? DW_BIAS_LO : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum);
      } else if (NCB == 4 && duty == DW_BIAS_HI) {
        asm volatile("; column sums, high half" ::: "memory");
        dw_stream<NCB, NCB == 4 ? DW_BIAS_HI : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum);
      } else if (NCB == 4 && (duty == DW_RAY_LO || duty == DW_RAY_HI)) {
        const RayDuty rd{p.raysum, p.ray_nc, p.ray_nf, p.rows_c, p.rows_c / (p.ray_nc > 0 ? p.ray_nc : 1)};
        if (duty == DW_RAY_LO) {
          asm volatile("; per-ray sums, low half" ::: "memory");
          dw_stream<NCB, NCB == 4 ? DW_RAY_LO : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane);
        } else {
          asm volatile("; per-ray sums, high half" ::: "memory");
          dw_stream<NCB, NCB == 4 ? DW_RAY_HI : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum, rd, lane);
        }
      } else if (NCB == 4 && duty == DW_SIG) {
        asm volatile("; sigma head" ::: "memory");
        dw_stream<NCB, NCB == 4 ? DW_SIG : DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum);
      } else {
        asm volatile("; no duty" ::: "memory");
        dw_stream<NCB, DW_PLAIN>(gbase, xbase, sbase, goff, xoff, soff, r_begin, r_end, acc, bsum);
      }
    }
    if (r_full < r_stop) dw_stream_tail<NCB>(gbase, xbase, sbase, goff, xoff, soff, h, r_full, r_stop, duty, acc, bsum);
  }
  constexpr size_t wave_rows = (size_t)4 * NCB * 16;
  float* sl = slabs + p.slab_off;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sl[dwi_row_off(p, (size_t)wv * wave_rows + (size_t)(ca * NCB + cb) * 16 + r, lw) + lane] = acc[ca][cb][r];
  // the duty's sums: 4 columns per lane (bias: columns 4q.. of the G block; sigma head: columns NCB q.. of the X block), rows h, h + 2, ...
#pragma unroll
  for (int c = 0; c < 4; ++c) bsum[c] += __shfl_xor(bsum[c], 32);
  if (MIXED) {
#pragma unroll
    for (int c = 0; c < 4; ++c) ssum[c] += __shfl_xor(ssum[c], 32);
  }
  if (h == 0) {
    float* bs = sl + dwi_sums_off(p, lw) + (size_t)wv * 128;
    float* sg = bs + 4 * 128;
    const bool is_sig = duty == DW_SIG;
#pragma unroll
    for (int c = 0; c < 4; ++c) bs[4 * q + c] = is_sig ? 0.f : bsum[c];
    if (p.has_sig) {
#pragma unroll
      for (int c = 0; c < 4; ++c) sg[4 * q + c] = MIXED ? ssum[c] : (is_sig ? bsum[c] : 0.f);
    }
  }
#ifdef NERF_STAMPS
  if (p.stamps && lane == 0) {  // per wave: 100 MHz timestamps, XCC and hardware ids
    unsigned long long* r = p.stamps + ((size_t)lw * 8 + wv) * 4;
    r[0] = t_start;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    r[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  }
#endif
}

template <int NCB, bool MIXED = false>
__global__ __launch_bounds__(256) void k_dw4(const DwItem p, const long long Mtot, float* __restrict__ slabs) {
  dw4_body<NCB, MIXED>(p, Mtot, slabs, blockIdx.x);
}

// The seven 256 x 256 products of a step (layers 1..7) in ONE launch, grid (workgroups per product, products): a product gets
// DW_GROUP_WGS = 36 workgroups (7 x 36 = 252 of the 256 CUs, one round), each on 1/36 of the rows.  With a launch per product and a
// workgroup per CU every product wrote and re-read 256 slabs of 256 KiB -- 0.9 GB per step whatever the batch: a quarter of this phase at
// 512 rays (the reference's own regime is BATCH_RAY = 400); now a seventh of that, and six launch boundaries fewer.  Measured
// (scripts/ab_variants.sh, phase time per step): 400 rays 0.92 -> 0.76 ms, 512: 1.07 -> 0.90, 1024: 1.79 -> 1.61, 4096: 6.15 -> 6.00.
// (Round 2's "one launch for all products" measured no gain: its ragged last workgroup ran the row-pair loop over its whole range --
// see r_full in dw4_body -- and hid the gain.)
__global__ __launch_bounds__(256) void k_dw4_group(const DwBatch b, const long long Mtot, float* __restrict__ slabs) {
  dw4_body<4, false>(b.item[blockIdx.y], Mtot, slabs, blockIdx.x);
}

// The other products: one launch each, one small kernel per block shape (this compiler schedules the stage loads of a loop
// differently when other loops share its kernel).
// Sums the slabs of every product of the step and scatters into the nn.Linear-layout gradients: grid (blocks, items).
// Weight blocks: 256 consecutive slab elements per block, thread = (4 consecutive elements, a quarter of the slabs) with four
// 16-byte loads in flight, the quarters combined through LDS in a fixed order (deterministic, no float atomics).  The blocks
// behind them do the column sums (bias gradients, sigma-head weights), one element per thread.
__global__ __launch_bounds__(256) void k_dw_reduce(const DwBatch b) {
  __shared__ float4 part[3][64];
  const DwItem& p = b.item[blockIdx.y];
  const int nblocks = dwi_nblocks(p), msubs = dwi_msubs(p), in_blocks = dwi_in_blocks(p);
  const size_t wave_floats = dwi_wave_floats(p);
  const int n_w = nblocks * (int)wave_floats;  // weight elements (padded), a multiple of 256
  const int n_b = p.thin ? 4 : (p.db ? p.nout : 0);
  const int n_s = p.has_sig ? p.nin : 0;
  const int wblocks = n_w / 256;
  const float* base = b.slabs + p.slab_off;
  const int waves = dwi_waves(p);
  if ((int)blockIdx.x < wblocks) {
    const int quad = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x * 64 + quad) * 4;
    const int blk = e / (int)wave_floats;
    int r = e - blk * (int)wave_floats;
    auto add4 = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
    const int kq = p.nwg / 4, k0 = kp * kq, k1 = kp == 3 ? p.nwg : k0 + kq;
    const int lane0 = r & 63;
    const size_t wgs = dwi_wg_stride(p);
    for (int ms = 0; ms < msubs; ++ms) {  // fixed order: msub, then workgroup, four partial sums
      const size_t row = ((size_t)(ms * nblocks + blk) * wave_floats + r) / 64;
      const float* q = base + dwi_row_off(p, row, 0) + lane0;
      int k = k0;
      for (; k + 4 <= k1; k += 4) {
        add4(t0, *reinterpret_cast<const float4*>(q + (size_t)k * wgs));
        add4(t1, *reinterpret_cast<const float4*>(q + (size_t)(k + 1) * wgs));
        add4(t2, *reinterpret_cast<const float4*>(q + (size_t)(k + 2) * wgs));
        add4(t3, *reinterpret_cast<const float4*>(q + (size_t)(k + 3) * wgs));
      }
      for (; k < k1; ++k) add4(t0, *reinterpret_cast<const float4*>(q + (size_t)k * wgs));
    }
    add4(t0, t1); add4(t2, t3); add4(t0, t2);
    if (kp) part[kp - 1][quad] = t0;
    __syncthreads();
    if (kp) return;
    add4(t0, part[0][quad]);
    float4 u = part[1][quad];
    add4(u, part[2][quad]);
    add4(t0, u);
    const float sv[4] = {t0.x, t0.y, t0.z, t0.w};
    r >>= 6;
    const int reg = r & 15; r >>= 4;
    const int ncb = dwi_ncb(p);
    const int cb = r % ncb, ca = r / ncb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lane = lane0 + j;
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);  // row of the 32 x 32 tile
      if (p.thin) {
        const int out = i;                                   // row of A: 0..2 colour (3: dsigma_pre, unused here)
        const int in = blk * 64 + 2 * (lane & 31) + cb;      // column of c
        if (out < 3) p.dW[(size_t)out * HALF + in] = sv[j];  // dW_color[3][128]
      } else {
        const int oA = (blk / in_blocks) * 128, iB = (blk % in_blocks) * 32 * ncb;
        const int out = oA + 4 * i + ca;
        const int in = iB + ncb * (lane & 31) + cb;
        if (in < p.nin_real) p.dW[(size_t)out * p.ldw + p.col0 + in] = sv[j];
      }
    }
    return;
  }
  // column sums: one wave per column, lane = workgroups k = lane, lane + 64, ...; combined by a fixed shuffle tree
  const int col = ((int)blockIdx.x - wblocks) * 4 + ((int)threadIdx.x >> 6), kp = (int)threadIdx.x & 63;
  if (col >= n_b + n_s) return;  // wave-uniform
  const size_t sums_per_wg = dwi_sums_per_wg(p);
  float s = 0.f;
  if (col < n_b) {
    const int o = col;  // column of G (thin: column of A)
    const float* bs = base + dwi_sums_off(p, 0);
    const int ob = p.thin ? 0 : o / 128, oi = p.thin ? 4 * o : o % 128;
    for (int w = 0; w < waves; ++w) {  // the waves that summed these columns, in wave order
      const int blk = w % nblocks;
      if (!dwi_sums_columns(dwi_duty(p, p.thin ? 0 : blk / in_blocks, blk % in_blocks)) || (!p.thin && blk / in_blocks != ob)) continue;
      const float* q = bs + (size_t)w * 128 + oi;
      for (int k = kp; k < p.nwg; k += 64) s += q[(size_t)k * sums_per_wg];
    }
  } else {
    const int o = col - n_b;  // column of X: sigma-head weight gradient
    const int ncb = dwi_ncb(p);
    const int bj = o / (32 * ncb), oi = o % (32 * ncb);
    const float* sg = base + dwi_sums_off(p, 0) + (size_t)waves * 128;
    for (int w = 0; w < waves; ++w) {
      const int blk = w % nblocks;
      if (blk % in_blocks != bj || !dwi_sums_sigma(dwi_duty(p, blk / in_blocks, bj))) continue;
      const float* q = sg + (size_t)w * 128 + oi;
      for (int k = kp; k < p.nwg; k += 64) s += q[(size_t)k * sums_per_wg];
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  if (kp) return;
  if (col < n_b) {
    if (p.thin) { if (col < 3) p.db[col] = s; else p.db2[0] = s; }  // db_color[3], db_sigma
    else p.db[col] = s;
  } else {
    p.dW2[col - n_b] = s;
  }
}

// The dir_info product may carry the per-ray sums when every wave's row range starts and ends on ray boundaries of its pass and no
// range is ragged (then the pipelined loop runs everywhere): ranges are multiples of both sample counts, the coarse pass's rows
// are a multiple of the fine pass's samples per ray, and the rows fill whole ranges.
bool dw_ray_duty_ok(const DwItem& p, long long Mtot, int B, int Nc, int Nf) {
  if (p.thin || dwi_ncb(p) != 4 || dwi_in_blocks(p) != 2 || !p.db || p.nout != 128) return false;
  if (Nc % (2 * DW4_DEPTH) || Nf % (2 * DW4_DEPTH) || Mtot != (long long)B * (Nc + Nf)) return false;
  const int msubs = dwi_msubs(p), gran = 2 * DW4_DEPTH * msubs, Mrows = (int)Mtot;
  const int per_wg = ((Mrows + p.nwg - 1) / p.nwg + gran - 1) / gran * gran, per_wave = per_wg / msubs;
  return per_wave % Nc == 0 && per_wave % Nf == 0 && ((long long)B * Nc) % Nf == 0 && Mrows % per_wave == 0;
}

size_t dw_item_slab_floats(const DwItem& p) { return (size_t)p.nwg * dwi_wg_floats(p); }

// point_info folded into dir_info (common.h SEG_FOLD): with M = sum_m dpre_dir[m] (x) h7[m] (128 x 256, the product above) and
// db_dir = sum_m dpre_dir[m], the gradients of the two ORIGINAL parameter tensors follow exactly (feat = W_pi h7 + b_pi):
//   dW_pi            = W_dir[:, 24:]^T M                  (256 x 256)      db_pi = W_dir[:, 24:]^T db_dir
//   dW_dir[:, 24:]   = M W_pi^T + db_dir (x) b_pi         (128 x 256)
// 16.8 M MACs per step, fp32 fma chains in a fixed order.  Blocks [0, 256): row i of dW_pi; [256, 384): row o of dW_dir; 384: db_pi.
__global__ __launch_bounds__(256) void k_fold_grads(const FoldGradArgs a) {
  __shared__ float mrow[WIDTH];
  const int t = threadIdx.x, b = blockIdx.x;
  constexpr int LD = WIDTH + DIR_DIM;
  if (b < WIDTH) {  // dW_pi[i][k] = sum_o W_dir[o][24 + i] * M[o][k]: the W_dir element is block-uniform, M coalesced over k = t
    const int i = b;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int o = 0; o < HALF; o += 2) {
      s0 = __builtin_fmaf(a.w_dir[(size_t)o * LD + DIR_DIM + i], a.M[(size_t)o * WIDTH + t], s0);
      s1 = __builtin_fmaf(a.w_dir[(size_t)(o + 1) * LD + DIR_DIM + i], a.M[(size_t)(o + 1) * WIDTH + t], s1);
    }
    a.dW_pi[(size_t)i * WIDTH + t] = s0 + s1;
  } else if (b < WIDTH + HALF) {  // dW_dir[o][24 + i] = sum_k M[o][k] * W_pi[i][k] + db_dir[o] * b_pi[i]: thread i walks ITS row of W_pi
    const int o = b - WIDTH, i = t;
    mrow[t] = a.M[(size_t)o * WIDTH + t];
    __syncthreads();
    const float4* wr = reinterpret_cast<const float4*>(a.w_pi + (size_t)i * WIDTH);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int k4 = 0; k4 < WIDTH / 4; ++k4) {
      const float4 q = wr[k4];
      s0 = __builtin_fmaf(mrow[4 * k4 + 0], q.x, s0);
      s1 = __builtin_fmaf(mrow[4 * k4 + 1], q.y, s1);
      s2 = __builtin_fmaf(mrow[4 * k4 + 2], q.z, s2);
      s3 = __builtin_fmaf(mrow[4 * k4 + 3], q.w, s3);
    }
    a.dW_dir[(size_t)o * LD + DIR_DIM + i] = __builtin_fmaf(a.db_dir[o], a.b_pi[i], (s0 + s1) + (s2 + s3));
  } else {  // db_pi[i] = sum_o W_dir[o][24 + i] * db_dir[o]
    const int i = t;
    float s = 0.f;
#pragma unroll 8
    for (int o = 0; o < HALF; ++o) s = __builtin_fmaf(a.w_dir[(size_t)o * LD + DIR_DIM + i], a.db_dir[o], s);
    a.db_pi[i] = s;
  }
}

hipError_t launch_fold_grads(const FoldGradArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_grads, dim3(WIDTH + HALF + 1), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_dw(const DwBatch& b, long long Mtot, float* slabs, hipStream_t st, int first, int count) {
  const int last = count < 0 ? b.n : first + count;
  if (b.grouped > 0 && first == 0) {  // (a range that starts inside the group is not something api.hip asks for)
    if (last < b.grouped) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dw4_group, dim3(b.item[0].nwg, b.grouped), dim3(256), 0, st, b, Mtot, slabs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    first = b.grouped;
  }
  for (int i = first; i < last; ++i) {
    const DwItem& p = b.item[i];
    if (p.thin) hipLaunchKernelGGL(k_dw_thin, dim3(p.nwg), dim3(512), 0, st, p, Mtot, slabs);
    else if (dwi_ncb(p) == 4 && p.has_sig && p.nout == 128) hipLaunchKernelGGL((k_dw4<4, true>), dim3(p.nwg), dim3(256), 0, st, p, Mtot, slabs);
    else if (dwi_ncb(p) == 4) hipLaunchKernelGGL((k_dw4<4>), dim3(p.nwg), dim3(256), 0, st, p, Mtot, slabs);
    else hipLaunchKernelGGL((k_dw4<2>), dim3(p.nwg), dim3(256), 0, st, p, Mtot, slabs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_dw_reduce(const DwBatch& b_in, hipStream_t st, int first, int count) {
  // items [first, first + count) (count < 0: all): the kernel indexes items with blockIdx.y, so the range is shifted to the front
  DwBatch b = b_in;
  const int last = count < 0 ? b_in.n : first + count;
  b.n = last - first;
  for (int i = 0; i < b.n; ++i) b.item[i] = b_in.item[first + i];
  int most = 0;
  for (int i = 0; i < b.n; ++i) {
    const DwItem& p = b.item[i];
    const int blocks = dwi_nblocks(p) * (int)dwi_wave_floats(p) / 256 + ((p.thin ? 4 : (p.db ? p.nout : 0)) + (p.has_sig ? p.nin : 0) + 3) / 4;
    most = blocks > most ? blocks : most;
  }
  hipLaunchKernelGGL(k_dw_reduce, dim3(most, b.n), dim3(256), 0, st, b);
  return hipGetLastError();
}

}  // namespace nerf
