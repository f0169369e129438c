// field_bwd.hip -- backward of the fused field query for MI355X (gfx950).
//
//  k_field_bwd<FINE>  the LDS-tile form of the dX chain (A/B reference; the product path is field_bwd_reg.hip): per 64-sample tile, walks the network backwards with the TRANSPOSED
//                     weights as the MFMA A operand (same LDS/accumulator layout as the forward kernel),
//                     applies the ReLU masks saved by the forward pass, and streams every layer's
//                     pre-activation gradient to HBM ([NGRAD][Mtot][256]) for the weight-gradient GEMMs.
//                     FINE also back-propagates into the encoding inputs: d loss/d gamma_p (skip layer +
//                     layer 0) -> d loss/d point -> d loss/d t_fine  (the reference does NOT detach t_fine,
//                     nerf.py:259 -- quirk Q9).
//  k_dw               dW[out][in] = sum_m G[m][out] * X[m][in] as a split-M fp32 MFMA GEMM: the reduction
//                     index is the sample, both operands are read straight from their row-major HBM images
//                     (lane l <- columns 4(l&31)..+3 of G and 2(l&31)..+1 of X, rows m, m+1), one 128 x 64 output
//                     block (128 accumulator VGPRs) per wave, 8 waves, one workgroup per CU, a branch-free
//                     3-stage register rotation with pinned prefetches, per-wave partial slabs summed by
//                     k_dw_reduce (deterministic; no float atomics).  Bias gradients (column sums of G) ride
//                     along on the VALU.
//  k_small_*          the thin heads: colour (3x128), sigma (1x256), direction encoding part of dir_info.
//
// Autograd spans replaced: backward of Network.forward (nerf.py:101-124), Encoder.forward (nerf.py:135-167)
// and the sample-point arithmetic (nerf.py:200-216) as invoked by loss.backward() at nerf.py:473.
#include "field_common.h"

namespace nerf {

// one 32x32 tile: acc += W^T tile(ft) [32 x 8KB] * act rows [st*32, st*32+32)
template <int KB>
__device__ __forceinline__ void mfma_tile1(const float4* __restrict__ wseg, int ft, int st, const float* act, int lane, f32x16& acc) {
  const int j = lane & 31, h = lane >> 5;
  const float* ap = act + (st * 32 + j) * LDA + 4 * h;
  const float4* wbase = wseg + (size_t)ft * KB * 64 + lane;
  float4 wc = wbase[0], wn;
#pragma unroll 2
  for (int kb = 0; kb < KB; ++kb) {
    const int kn = (kb + 1 < KB) ? kb + 1 : kb;
    wn = wbase[(size_t)kn * 64];
    const float4 a0 = *reinterpret_cast<const float4*>(ap + kb * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, a0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, a0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, a0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, a0.w, acc, 0, 0, 0);
    wc = wn;
  }
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][st][r] = 0.f;
}

// accumulators -> LDS rows, zeroing the entries whose forward activation was not > 0
__device__ __forceinline__ void acc_store_masked(float* act, int fbase, int lane, const f32x16 (&acc)[2][2], const uint16_t* mask) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const unsigned bits = mask[(f * 2 + st) * 256];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = (bits >> (4 * g + 0)) & 1u ? acc[f][st][4 * g + 0] : 0.f;
        v.y = (bits >> (4 * g + 1)) & 1u ? acc[f][st][4 * g + 1] : 0.f;
        v.z = (bits >> (4 * g + 2)) & 1u ? acc[f][st][4 * g + 2] : 0.f;
        v.w = (bits >> (4 * g + 3)) & 1u ? acc[f][st][4 * g + 3] : 0.f;
        *reinterpret_cast<float4*>(act + (st * 32 + j) * LDA + fbase + f * 32 + 8 * g + 4 * h) = v;
      }
    }
}

template <bool FINE>
__global__ __launch_bounds__(256, 2) void k_field_bwd(const FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;
  float* scr = smem + TM * LDA;  // [4][64][3]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * TM;
  const int sm = lane;
  const int m = m0 + sm;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int nrows = (a.M - m0) < TM ? (a.M - m0) : TM;
  const long long grow0 = (long long)a.row0 + m0;
  const size_t MS = (size_t)a.Mtot * WIDTH;
  const uint16_t* mk = a.masks + ((size_t)(a.tile0 + blockIdx.x) * 4) * 256 + tid;
  const size_t MKS = (size_t)a.tiles_tot * 4 * 256;
  const int fbase = wv * 64;

  // ---- colour head backward (VALU): rgb = sigmoid(z), z = W_c c + b; c = relu(pre_d)
  {
    float dz[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float o = a.rgb[(size_t)mc * 3 + ch];
      dz[ch] = valid ? a.drgb[(size_t)mc * 3 + ch] * ((1.0f - o) * o) : 0.f;
    }
    if (wv == 0 && valid) {
      float4 v = make_float4(dz[0], dz[1], dz[2], 0.f);
      *reinterpret_cast<float4*>(a.dz + (size_t)(a.row0 + m) * 4) = v;
    }
    const float* crow = a.save + S_C * MS + (size_t)(a.row0 + mc) * WIDTH + wv * 32;
    const float* wc = a.w.p[W_COLOR] + wv * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 cv = *reinterpret_cast<const float4*>(crow + k);
      float4 d;
      d.x = cv.x > 0.f ? __builtin_fmaf(wc[2 * HALF + k + 0], dz[2], __builtin_fmaf(wc[HALF + k + 0], dz[1], wc[k + 0] * dz[0])) : 0.f;
      d.y = cv.y > 0.f ? __builtin_fmaf(wc[2 * HALF + k + 1], dz[2], __builtin_fmaf(wc[HALF + k + 1], dz[1], wc[k + 1] * dz[0])) : 0.f;
      d.z = cv.z > 0.f ? __builtin_fmaf(wc[2 * HALF + k + 2], dz[2], __builtin_fmaf(wc[HALF + k + 2], dz[1], wc[k + 2] * dz[0])) : 0.f;
      d.w = cv.w > 0.f ? __builtin_fmaf(wc[2 * HALF + k + 3], dz[2], __builtin_fmaf(wc[HALF + k + 3], dz[1], wc[k + 3] * dz[0])) : 0.f;
      *reinterpret_cast<float4*>(act + sm * LDA + wv * 32 + k) = d;
    }
  }
  __syncthreads();
  save_rows(act, a.G + G_D * MS, grow0, nrows, 32, WIDTH, tid);

  f32x16 acc[2][2];
  // ---- dir_info backward: dfeat = W_d[:, 24:]^T dpre_d   (256 <- 128)
  acc_zero(acc);
  mfma_layer<16, 2>(a.wp + seg_off4(SEG_T_DIR), wv * 2, act, 0, lane, acc);
  __syncthreads();
  acc_store<2, false>(act, fbase, lane, acc);
  __syncthreads();
  save_rows(act, a.G + G_PI * MS, grow0, nrows, 64, WIDTH, tid);

  // ---- point_info backward + sigma head: dh7 = W_pi^T dfeat + w_sigma * dsigma_pre, masked by h7 > 0
  {
    const float sp = a.spre[a.row0 + mc];
    const float sgn = sp > 0.f ? 1.0f : (sp < 0.f ? -1.0f : 0.f);  // d|x|/dx with sign(0) = 0 like torch
    const float ds = valid ? a.dsig[mc] * sgn : 0.f;
    if (wv == 0 && valid) a.dspre[a.row0 + m] = ds;
    const int j = lane & 31, h = lane >> 5;
    const float ds0 = __shfl(ds, j), ds1 = __shfl(ds, j + 32);
    const float* ws = a.w.p[W_SIGMA];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 wq = *reinterpret_cast<const float4*>(ws + fbase + f * 32 + 8 * g + 4 * h);
        acc[f][0][4 * g + 0] = wq.x * ds0; acc[f][0][4 * g + 1] = wq.y * ds0; acc[f][0][4 * g + 2] = wq.z * ds0; acc[f][0][4 * g + 3] = wq.w * ds0;
        acc[f][1][4 * g + 0] = wq.x * ds1; acc[f][1][4 * g + 1] = wq.y * ds1; acc[f][1][4 * g + 2] = wq.z * ds1; acc[f][1][4 * g + 3] = wq.w * ds1;
      }
  }
  mfma_layer<32, 2>(a.wp + seg_off4(SEG_T_PI), wv * 2, act, 0, lane, acc);
  __syncthreads();
  acc_store_masked(act, fbase, lane, acc, mk + 7 * MKS);
  __syncthreads();
  save_rows(act, a.G + 7 * MS, grow0, nrows, 64, WIDTH, tid);

  // ---- layers 7, 6, 5: dpre_{i-1} = (W_i^T dpre_i) masked by h_{i-1} > 0
#pragma unroll 1
  for (int i = 7; i >= 5; --i) {
    acc_zero(acc);
    mfma_layer<32, 2>(a.wp + seg_off4(SEG_T_L7) + (size_t)(7 - i) * 8 * 32 * 64, wv * 2, act, 0, lane, acc);
    __syncthreads();
    acc_store_masked(act, fbase, lane, acc, mk + (size_t)(i - 1) * MKS);
    __syncthreads();
    save_rows(act, a.G + (size_t)(i - 1) * MS, grow0, nrows, 64, WIDTH, tid);
  }

  // ---- layer 4 (input cat(h3, gamma_p)): dh3 and, for the fine pass, d gamma_p through the skip connection
  f32x16 accg;
#pragma unroll
  for (int r = 0; r < 16; ++r) accg[r] = 0.f;
  acc_zero(acc);
  mfma_layer<32, 2>(a.wp + seg_off4(SEG_T_L4A), wv * 2, act, 0, lane, acc);
  if (FINE) mfma_tile1<32>(a.wp + seg_off4(SEG_T_L4B), wv >> 1, wv & 1, act, lane, accg);
  __syncthreads();
  acc_store_masked(act, fbase, lane, acc, mk + 3 * MKS);
  __syncthreads();
  save_rows(act, a.G + 3 * MS, grow0, nrows, 64, WIDTH, tid);

  // ---- layers 3, 2, 1
#pragma unroll 1
  for (int i = 3; i >= 1; --i) {
    acc_zero(acc);
    mfma_layer<32, 2>(a.wp + seg_off4(SEG_T_L3) + (size_t)(3 - i) * 8 * 32 * 64, wv * 2, act, 0, lane, acc);
    __syncthreads();
    acc_store_masked(act, fbase, lane, acc, mk + (size_t)(i - 1) * MKS);
    __syncthreads();
    save_rows(act, a.G + (size_t)(i - 1) * MS, grow0, nrows, 64, WIDTH, tid);
  }

  if (FINE) {
    // ---- layer 0: d gamma_p += W_0^T dpre_0; then gamma -> point -> depth
    mfma_tile1<32>(a.wp + seg_off4(SEG_T_L0), wv >> 1, wv & 1, act, lane, accg);
    __syncthreads();
    {
      const int j = lane & 31, h = lane >> 5;
      const int ft = wv >> 1, st = wv & 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(accg[4 * g], accg[4 * g + 1], accg[4 * g + 2], accg[4 * g + 3]);
        *reinterpret_cast<float4*>(act + (st * 32 + j) * LDA + ft * 32 + 8 * g + 4 * h) = v;
      }
    }
    __syncthreads();
    const int ray = mc / a.N;
    const float* rf = a.rayf + (size_t)ray * RAYF;
    float p[3];
    sample_point(rf, a.t[mc], p);
    float dp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = wv * 8 + i;  // wave-uniform (c, l) pair
      if (e < 30) {
        const int c = e / 10, l = e - 10 * c;
        const float x = (c == 0) ? p[0] : ((c == 1) ? p[1] : p[2]);
        const float fl = __uint_as_float(kFreqPointBits[l]);
        const float ph = x * fl;
        const float2 dg = *reinterpret_cast<const float2*>(act + sm * LDA + c * 20 + 2 * l);
        // d/dx [sin(f x), cos(f x)] . dgamma = f (cos * dg_sin - sin * dg_cos)
        float sn, cn;
        sincos_phase(ph, sn, cn);
        const float dph = cn * dg.x - sn * dg.y;
        const float contrib = fl * dph;
        if (c == 0) dp[0] += contrib; else if (c == 1) dp[1] += contrib; else dp[2] += contrib;
      }
    }
    scr[(wv * 64 + sm) * 3 + 0] = dp[0];
    scr[(wv * 64 + sm) * 3 + 1] = dp[1];
    scr[(wv * 64 + sm) * 3 + 2] = dp[2];
    __syncthreads();
    if (wv == 0 && valid) {
      float d3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) d3[c] = (scr[sm * 3 + c] + scr[(64 + sm) * 3 + c]) + (scr[(128 + sm) * 3 + c] + scr[(192 + sm) * 3 + c]);
      // point = R (d_cam t) + o  =>  d point / d t = R d_cam = d_wrd
      const float dtp = __builtin_fmaf(rf[RF_DWRD + 2], d3[2], __builtin_fmaf(rf[RF_DWRD + 1], d3[1], rf[RF_DWRD] * d3[0]));
      a.dt[m] += dtp;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient GEMM
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
__device__ __forceinline__ float comp(const float2& v, int c) { return c == 0 ? v.x : v.y; }

constexpr int DW_UNROLL = 4;   // k-steps (row pairs) per pipeline stage
constexpr int DW_STAGES = 3;   // register stages in flight (2 prefetched ahead of the one being multiplied)
constexpr int DW_CB = 2;       // B components per lane: a wave owns a 128 (out) x 64 (in) block
constexpr int DW_WAVES = 8;    // 512 threads, two waves per SIMD

struct DwStage {
  float4 a[DW_UNROLL];
  float2 b[DW_UNROLL];
};

// main loop of k_dw over rows [r_begin, r_end).  CHECK = false: every row of every stage is in range (no selects
// between the loads and their use, so the loads of two stages stay in flight behind the MFMAs of the third).
template <bool CHECK>
__device__ __forceinline__ void dw_rows(const DwProblem& p, const float* __restrict__ gp, const float* __restrict__ xp, long long r_begin,
                                        long long r_end, int h, bool do_bias, f32x16 (&acc)[4][DW_CB], float (&bsum)[4]) {
  auto load = [&](long long r0, DwStage& S) {
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) {
      // both operands have row stride WIDTH floats (all buffers of the workspace do)
      if (CHECK) {
        long long r = r0 + 2 * u + h;
        r = r < r_end ? r : r_end - 1;
        S.a[u] = *reinterpret_cast<const float4*>(gp + (size_t)r * WIDTH);
        S.b[u] = *reinterpret_cast<const float2*>(xp + (size_t)r * WIDTH);
      } else {
        const size_t off = (size_t)(r0 + h) * WIDTH + (size_t)u * 2 * WIDTH;
        S.a[u] = *reinterpret_cast<const float4*>(gp + off);
        S.b[u] = *reinterpret_cast<const float2*>(xp + off);
      }
    }
  };
  auto mul = [&](long long r0, const DwStage& S) {
#pragma unroll
    for (int u = 0; u < DW_UNROLL; ++u) {
      float4 a = S.a[u];
      if (CHECK) {
        if (r0 + 2 * u + h >= r_end) a = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the end contribute nothing
      }
#pragma unroll
      for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < DW_CB; ++cb) acc[ca][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a, ca), comp(S.b[u], cb), acc[ca][cb], 0, 0, 0);
      if (do_bias) {
        bsum[0] += a.x; bsum[1] += a.y; bsum[2] += a.z; bsum[3] += a.w;
      }
    }
  };
  constexpr long long G = 2 * DW_UNROLL;  // rows per stage
  DwStage s0, s1, s2;
  // branch-free rotation: the two prefetches past the end re-read the last stage (valid memory, never multiplied)
  const long long r_last = CHECK ? r_end : r_end - G;
  auto at = [&](long long r) { return (CHECK || r <= r_last) ? r : r_last; };
  load(r_begin, s0);
  load(at(r_begin + G), s1);
  for (long long r0 = r_begin; r0 < r_end; r0 += 3 * G) {
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

// Slab layout per wave block: [cA 4][cB 2][reg 16][lane 64] floats; after the nout*nin block values come nout column sums.
__global__ __launch_bounds__(512, 2) void k_dw(const DwProblem p) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int in_blocks = p.nin / 64;
  const int nblocks = (p.nout / 128) * in_blocks;  // 8, 4 or 2
  const int msubs = DW_WAVES / nblocks;
  const int blk = wv % nblocks, msub = wv / nblocks;
  const int oA = (blk / in_blocks) * 128, iB = (blk % in_blocks) * 64;
  const long long gran = (long long)2 * DW_UNROLL * DW_STAGES * msubs;
  const long long per_wg = ((p.Mtot + DW_WGS - 1) / DW_WGS + gran - 1) / gran * gran;
  const long long per_wave = per_wg / msubs;  // a multiple of the 24 rows of one pipeline round
  const long long r_begin = (long long)blockIdx.x * per_wg + (long long)msub * per_wave;
  const long long r_nom = r_begin + per_wave;
  const long long r_end = r_nom > p.Mtot ? p.Mtot : r_nom;
  const int h = lane >> 5, q = lane & 31;

  f32x16 acc[4][DW_CB];
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < DW_CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (p.db != nullptr) && (iB == 0);

  const float* gp = p.G + oA + 4 * q;
  const float* xp = p.X + iB + DW_CB * q;
  if (r_begin < r_end) {
    if (r_nom <= p.Mtot)
      dw_rows<false>(p, gp, xp, r_begin, r_end, h, do_bias, acc, bsum);
    else
      dw_rows<true>(p, gp, xp, r_begin, r_end, h, do_bias, acc, bsum);
  }
  // write this wave's slab
  const size_t slab_floats = (size_t)p.nout * p.nin + p.nout;
  constexpr size_t wave_floats = (size_t)4 * DW_CB * 16 * 64;  // 128 x 64
  float* slab = p.slabs + ((size_t)blockIdx.x * msubs + msub) * slab_floats;
  float* ws = slab + (size_t)blk * wave_floats;
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < DW_CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[((size_t)(ca * DW_CB + cb) * 16 + r) * 64 + lane] = acc[ca][cb][r];
  if (do_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] += __shfl_xor(bsum[c], 32);
    if (h == 0) {
      float* bs = slab + (size_t)p.nout * p.nin + oA;
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[4 * q + c] = bsum[c];
    }
  }
}

// sums the slabs and scatters into the nn.Linear-layout gradient
__global__ __launch_bounds__(256) void k_dw_reduce(const DwProblem p, int nslabs) {
  constexpr int CB = DW_CB;
  const int total = p.nout * p.nin + (p.db ? p.nout : 0);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const size_t slab_floats = (size_t)p.nout * p.nin + p.nout;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // nslabs is a multiple of 8: eight independent loads in flight
  for (int k = 0; k < nslabs; k += 8) {
    const float* q = p.slabs + (size_t)k * slab_floats + e;
    const float v0 = q[0], v1 = q[slab_floats], v2 = q[2 * slab_floats], v3 = q[3 * slab_floats];
    const float v4 = q[4 * slab_floats], v5 = q[5 * slab_floats], v6 = q[6 * slab_floats], v7 = q[7 * slab_floats];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    s0 += v4; s1 += v5; s2 += v6; s3 += v7;
  }
  const float s = (s0 + s1) + (s2 + s3);
  if (e < p.nout * p.nin) {
    const int wave_floats = 4 * CB * 16 * 64;
    const int in_blocks = p.nin / (32 * CB);
    const int blk = e / wave_floats;
    int r = e - blk * wave_floats;
    const int lane = r & 63; r >>= 6;
    const int reg = r & 15; r >>= 4;
    const int cb = r % CB, ca = r / CB;
    const int oA = (blk / in_blocks) * 128, iB = (blk % in_blocks) * (32 * CB);
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int out = oA + 4 * i + ca;
    const int in = iB + CB * (lane & 31) + cb;
    if (in < p.nin_real) p.dW[(size_t)out * p.ldw + p.col0 + in] = s;
  } else {
    p.db[e - p.nout * p.nin] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// thin heads
// ------------------------------------------------------------------------------------------------
// colour head (dW_c[3][128], db_c[3]) and sigma head (dw_sigma[256], db_sigma): plain VALU column reductions with
// one float atomic per (workgroup, output).  Destinations must be zeroed beforehand.
constexpr int SG_WGS = 4096;
__global__ __launch_bounds__(256) void k_small_heads(const SmallGradArgs a) {
  const int t = threadIdx.x;
  const long long per = (a.Mtot + SG_WGS - 1) / SG_WGS;
  const long long r0 = (long long)blockIdx.x * per;
  long long r1 = r0 + per;
  if (r1 > a.Mtot) r1 = a.Mtot;
  const size_t MS = (size_t)a.Mtot * WIDTH;
  const float* h7 = a.save + 7 * MS;
  const float* cc = a.save + S_C * MS;
  float as = 0.f, ac0 = 0.f, ac1 = 0.f, ac2 = 0.f, bs = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
  for (long long r = r0; r < r1; ++r) {
    const float ds = a.dspre[r];
    const float4 dz = *reinterpret_cast<const float4*>(a.dz + (size_t)r * 4);
    as = __builtin_fmaf(ds, h7[(size_t)r * WIDTH + t], as);
    if (t < HALF) {
      const float c = cc[(size_t)r * WIDTH + t];
      ac0 = __builtin_fmaf(dz.x, c, ac0);
      ac1 = __builtin_fmaf(dz.y, c, ac1);
      ac2 = __builtin_fmaf(dz.z, c, ac2);
    }
    if (t == 0) { bs += ds; b0 += dz.x; b1 += dz.y; b2 += dz.z; }
  }
  atomicAdd(a.dw_sigma + t, as);
  if (t < HALF) {
    atomicAdd(a.dW_color + t, ac0);
    atomicAdd(a.dW_color + HALF + t, ac1);
    atomicAdd(a.dW_color + 2 * HALF + t, ac2);
  }
  if (t == 0) {
    atomicAdd(a.db_sigma, bs);
    atomicAdd(a.db_color + 0, b0);
    atomicAdd(a.db_color + 1, b1);
    atomicAdd(a.db_color + 2, b2);
  }
}

// direction-encoding columns of dir_info: dW_d[o][k<24] = sum_ray gamma_d[ray][k] * sum_{samples of ray} dpre_d[m][o]
__global__ __launch_bounds__(128) void k_dir_ray_sums(const SmallGradArgs a) {
  const int ray = blockIdx.x, t = threadIdx.x;
  const size_t MS = (size_t)a.Mtot * WIDTH;
  const float* gd = a.G + G_D * MS;
  float s = 0.f;
  const size_t c0 = (size_t)ray * a.Nc, f0 = (size_t)a.B * a.Nc + (size_t)ray * a.Nf;
  for (int i = 0; i < a.Nc; ++i) s += gd[(c0 + i) * WIDTH + t];
  for (int i = 0; i < a.Nf; ++i) s += gd[(f0 + i) * WIDTH + t];
  a.sbuf[(size_t)ray * HALF + t] = s;
  if (t < 12) {
    const int c = t >> 2, l = t & 3;
    const float ph = a.rayf[(size_t)ray * RAYF + RF_DWRD + c] * __uint_as_float(kFreqDirBits[l]);
    a.gdbuf[(size_t)ray * DIR_DIM + c * 8 + 2 * l] = sinf(ph);
    a.gdbuf[(size_t)ray * DIR_DIM + c * 8 + 2 * l + 1] = cosf(ph);
  }
}
constexpr int DG_CHUNKS = 64;
__global__ __launch_bounds__(128) void k_dir_gamma_zero(const SmallGradArgs a) {
  const int o = threadIdx.x;
  for (int k = 0; k < DIR_DIM; ++k) a.dW_dir[(size_t)o * (WIDTH + DIR_DIM) + k] = 0.f;
}
__global__ __launch_bounds__(128) void k_dir_gamma_dw(const SmallGradArgs a) {
  // block (k, chunk of rays), thread o (0..127); one float atomic per (block, o)
  const int k = blockIdx.x, o = threadIdx.x;
  const int per = (a.B + DG_CHUNKS - 1) / DG_CHUNKS;
  const int r0 = blockIdx.y * per;
  const int r1 = (r0 + per) < a.B ? (r0 + per) : a.B;
  float s = 0.f;
  for (int ray = r0; ray < r1; ++ray) s = __builtin_fmaf(a.sbuf[(size_t)ray * HALF + o], a.gdbuf[(size_t)ray * DIR_DIM + k], s);
  if (r0 < r1) atomicAdd(a.dW_dir + (size_t)o * (WIDTH + DIR_DIM) + k, s);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
hipError_t launch_field_bwd(const FieldBwdArgs& a, bool fine, hipStream_t st) {
  const int tiles = (a.M + TM - 1) / TM;
  const size_t lds = FIELD_LDS_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> opted{0};
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(k_field_bwd<false>), reinterpret_cast<const void*>(k_field_bwd<true>)}, (int)lds)) return e;
  if (fine)
    hipLaunchKernelGGL(k_field_bwd<true>, dim3(tiles), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(k_field_bwd<false>, dim3(tiles), dim3(256), lds, st, a);
  return hipGetLastError();
}

size_t dw_slab_floats(int nout, int nin) {
  const int nblocks = (nout / 128) * (nin / 64);
  return (size_t)DW_WGS * (DW_WAVES / nblocks) * ((size_t)nout * nin + nout);
}
size_t dw_slab_floats_max() {
  size_t a = dw_slab_floats(256, 256), b = dw_slab_floats(128, 256), c = dw_slab_floats(256, 64);
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

hipError_t launch_dw(const DwProblem& p, hipStream_t st) {
  const int nblocks = (p.nout / 128) * (p.nin / 64);
  const int msubs = DW_WAVES / nblocks;
  hipLaunchKernelGGL(k_dw, dim3(DW_WGS), dim3(512), 0, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int total = p.nout * p.nin + (p.db ? p.nout : 0);
  hipLaunchKernelGGL(k_dw_reduce, dim3((total + 255) / 256), dim3(256), 0, st, p, DW_WGS * msubs);
  return hipGetLastError();
}

hipError_t launch_small_grads(const SmallGradArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_small_heads, dim3(SG_WGS), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_dir_ray_sums, dim3(a.B), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_dir_gamma_zero, dim3(1), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_dir_gamma_dw, dim3(DIR_DIM, DG_CHUNKS), dim3(128), 0, st, a);
  return hipGetLastError();
}

}  // namespace nerf
