// field_bwd.hip -- backward of the fused field query for MI355X (gfx950).
//
//  k_field_bwd<FINE>  the LDS-tile form of the dX chain (A/B reference; the product path is field_bwd_reg.hip): per 64-sample tile, walks the network backwards with the TRANSPOSED
//                     weights as the MFMA A operand (same LDS/accumulator layout as the forward kernel),
//                     applies the ReLU masks saved by the forward pass, and streams every layer's
//                     pre-activation gradient to HBM ([NGRAD][Mtot][256]) for the weight-gradient GEMMs.
//                     FINE also back-propagates into the encoding inputs: d loss/d gamma_p (skip layer +
//                     layer 0) -> d loss/d point -> d loss/d t_fine  (the reference does NOT detach t_fine,
//                     nerf.py:259 -- quirk Q9).
//  k_dir_*            the direction-encoding columns of dir_info: sum over rays of gamma_d (x) (per-ray sums of dpre_dir), summed
//                     in a fixed order (no float atomics); the per-ray sums normally come out of the dir_info product (dw_f32.hip).
//  (the weight-gradient GEMMs incl. the colour / sigma heads: dw_f32.hip)
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
  const size_t MS = (size_t)a.MSrows * WIDTH;
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

  // ---- point_info backward + sigma head: dh7 = W_pi^T dfeat + w_sigma * dsigma_pre, masked by h7 > 0
  {
    const float sp = a.spre[a.row0 + mc];
    const float sgn = sp > 0.f ? 1.0f : (sp < 0.f ? -1.0f : 0.f);  // d|x|/dx with sign(0) = 0 like torch
    const float ds = valid ? a.dsig[mc] * sgn : 0.f;
    if (wv == 0 && valid) {
      a.dspre[a.row0 + m] = ds;
      a.dz[(size_t)(a.row0 + m) * 4 + 3] = ds;  // (dz_r, dz_g, dz_b, dsigma_pre): A operand of the thin-heads product (dw_f32.hip)
    }
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
// direction-encoding columns of dir_info (the weight-gradient GEMMs and the thin heads live in dw_f32.hip)
// ------------------------------------------------------------------------------------------------
// direction-encoding columns of dir_info: dW_d[o][k<24] = sum_ray gamma_d[ray][k] * sum_{samples of ray} dpre_d[m][o]
// per-ray sums (fallback: ray counts / sample counts for which the dir_info product cannot carry the sums, dw_ray_duty_ok)
__global__ __launch_bounds__(512) void k_dir_ray_sums(const SmallGradArgs a) {
  // one ray per block: thread (part, column); the four parts take every fourth row, four partial sums each (fixed order)
  __shared__ float part_sum[4][HALF];
  const int ray = blockIdx.x, t = threadIdx.x & (HALF - 1), part = threadIdx.x >> 7;
  const size_t MS = (size_t)a.MSrows * WIDTH;
  const float* gd = a.G + G_D * MS;
  const size_t c0 = (size_t)ray * a.Nc, f0 = (size_t)a.B * a.Nc + (size_t)ray * a.Nf;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  auto span = [&](size_t r0, int n) {
    int i = part;
    for (; i + 12 < n; i += 16) {
      s0 += gd[(r0 + i) * WIDTH + t]; s1 += gd[(r0 + i + 4) * WIDTH + t];
      s2 += gd[(r0 + i + 8) * WIDTH + t]; s3 += gd[(r0 + i + 12) * WIDTH + t];
    }
    for (; i < n; i += 4) s0 += gd[(r0 + i) * WIDTH + t];
  };
  span(c0, a.Nc);
  span(f0, a.Nf);
  part_sum[part][t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0) {
    a.sbuf[(size_t)ray * HALF + t] = (part_sum[0][t] + part_sum[1][t]) + (part_sum[2][t] + part_sum[3][t]);
    a.sbuf[((size_t)a.B + ray) * HALF + t] = 0.f;  // (the sum over both passes sits in the coarse half)
  }
}
constexpr int DG_CHUNKS = 128;
// gamma_d of every ray (nerf.py:292-296 on the normalised direction)
__global__ __launch_bounds__(128) void k_dir_prep(const SmallGradArgs a) {
  const int i = blockIdx.x * 128 + threadIdx.x;  // (ray, pair)
  if (i < a.B * 12) {
    const int ray = i / 12, t = i - ray * 12, c = t >> 2, l = t & 3;
    const float ph = a.rayf[(size_t)ray * RAYF + RF_DWRD + c] * __uint_as_float(kFreqDirBits[l]);
    a.gdbuf[(size_t)ray * DIR_DIM + c * 8 + 2 * l] = sinf(ph);
    a.gdbuf[(size_t)ray * DIR_DIM + c * 8 + 2 * l + 1] = cosf(ph);
  }
}
// dW_dir[o][k < 24] = sum_ray gamma_d[ray][k] * (sum over the ray's samples of dpre_dir[.][o]) in two deterministic steps (no float
// atomics): block = a chunk of rays, thread o keeps the 24 partial sums of its output row; then one thread per (o, k) adds the
// chunks' partials in a fixed order.  `part` = [DG_CHUNKS][128][24] floats of scratch (the slab buffer, free by now).
__global__ __launch_bounds__(128) void k_dir_gamma_part(const SmallGradArgs a, float* __restrict__ part) {
  const int o = threadIdx.x;
  const int per = (a.B + DG_CHUNKS - 1) / DG_CHUNKS;
  const int r0 = blockIdx.x * per;
  const int r1 = (r0 + per) < a.B ? (r0 + per) : a.B;
  const float* sf = a.sbuf + (size_t)a.B * HALF;  // sums over the fine pass's rows
  float acc[DIR_DIM];
#pragma unroll
  for (int k = 0; k < DIR_DIM; ++k) acc[k] = 0.f;
  for (int ray = r0; ray < r1; ++ray) {
    const float s = a.sbuf[(size_t)ray * HALF + o] + sf[(size_t)ray * HALF + o];
    const float4* gd = reinterpret_cast<const float4*>(a.gdbuf + (size_t)ray * DIR_DIM);  // the same 96 bytes for every thread
#pragma unroll
    for (int k4 = 0; k4 < DIR_DIM / 4; ++k4) {
      const float4 g = gd[k4];
      acc[4 * k4 + 0] = __builtin_fmaf(s, g.x, acc[4 * k4 + 0]);
      acc[4 * k4 + 1] = __builtin_fmaf(s, g.y, acc[4 * k4 + 1]);
      acc[4 * k4 + 2] = __builtin_fmaf(s, g.z, acc[4 * k4 + 2]);
      acc[4 * k4 + 3] = __builtin_fmaf(s, g.w, acc[4 * k4 + 3]);
    }
  }
  float4* out = reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * HALF + o) * DIR_DIM);
#pragma unroll
  for (int k4 = 0; k4 < DIR_DIM / 4; ++k4) out[k4] = make_float4(acc[4 * k4], acc[4 * k4 + 1], acc[4 * k4 + 2], acc[4 * k4 + 3]);
}
__global__ __launch_bounds__(256) void k_dir_gamma_final(const SmallGradArgs a, const float* __restrict__ part) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // (o, k)
  if (e >= HALF * DIR_DIM) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
  for (int c = 0; c < DG_CHUNKS; c += 4) {
    s0 += part[(size_t)c * HALF * DIR_DIM + e];
    s1 += part[(size_t)(c + 1) * HALF * DIR_DIM + e];
    s2 += part[(size_t)(c + 2) * HALF * DIR_DIM + e];
    s3 += part[(size_t)(c + 3) * HALF * DIR_DIM + e];
  }
  const int o = e / DIR_DIM, k = e - o * DIR_DIM;
  a.dW_dir[(size_t)o * (WIDTH + DIR_DIM) + k] = (s0 + s1) + (s2 + s3);
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

hipError_t launch_small_grads(const SmallGradArgs& a, float* scratch, hipStream_t st) {
  if (!a.sums_done) hipLaunchKernelGGL(k_dir_ray_sums, dim3(a.B), dim3(512), 0, st, a);
  hipLaunchKernelGGL(k_dir_prep, dim3((a.B * 12 + 127) / 128), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_dir_gamma_part, dim3(DG_CHUNKS), dim3(128), 0, st, a, scratch);
  hipLaunchKernelGGL(k_dir_gamma_final, dim3((HALF * DIR_DIM + 255) / 256), dim3(256), 0, st, a, scratch);
  return hipGetLastError();
}

}  // namespace nerf
