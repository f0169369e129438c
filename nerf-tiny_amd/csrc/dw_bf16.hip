// dw_bf16.hip -- weight gradients of the bf16-MLP variant: dW[o][i] = sum over samples of G[s][o] * X[s][i] on
// v_mfma_f32_32x32x16_bf16 with the SAMPLE index as the MFMA k (MI355X / gfx950).
//
// G (pre-activation gradients, written by field_bwd_bf16.hip) and X (layer inputs, saved by field_fwd_bf16.hip) live in
// HBM in fragment layout: per wave block of 32 samples, 1-KiB pieces in which lane (sample j, half h) holds 8 features of
// its sample.  Both MFMA operands need the transpose (8 consecutive SAMPLES of one feature per lane), which gfx950's
// ds_read_b64_tr_b16 does on the way out of LDS:
//   * a workgroup (8 waves, one per 32 output rows o) walks its share of the wave blocks; block b's G and X pieces are
//     brought to a 4-slot LDS ring by direct-to-LDS loads three blocks ahead (one barrier per block);
//   * the LDS image is the fragment layout with the sample index XOR-swizzled per (piece parity, half) -- done on the
//     SOURCE address of the load, the LDS side of a direct load being lane-linear -- so that the 32 lanes of a half hit
//     64 different banks in every transposed read (unswizzled: 4-way conflicts);
//   * per block and wave: 2 k-steps x (1 A fragment + NIT B fragments), 2 transposed reads each, NIT (+1: bias
//     gradient = G^T . ones) MFMAs per k-step; accumulators (NIT x 16 fp32 registers) persist over the whole share;
//   * the per-workgroup partial sums go to a slab and are added up in workgroup order by k_dw_bf16_reduce: the result
//     does not depend on timing.
// The kernel is HBM-bound by construction (1 KiB of operands per 128 kFLOP): it is paced by the ring, not by the MFMAs.
#include "bf16_stream.h"

namespace nerf {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_p;

constexpr int DWB_NSLOT = 4;
constexpr int DWB_SLOT_BYTES = 32 * 1024;  // G (<= 16 KiB) + X (<= 16 KiB) of one wave block
constexpr int DWB_LDS_BYTES = DWB_NSLOT * DWB_SLOT_BYTES;
constexpr int DWB_WGS = 256;

struct DwBfArgs {
  const unsigned char* G;  // start of the gradient tensor (fragment layout), g_ks pieces per wave block
  const unsigned char* X;  // start of the input tensor, x_ks = 2 * NIT pieces per wave block
  int g_ks, o_tiles;       // o_tiles = output row tiles (waves w >= o_tiles only help loading)
  int wb_tot;
  float* slabs;            // [gridDim.x][o_tiles*32][NIT*32 + 1]  (last column: sum of G over the samples)
};

// LDS unit (16 bytes) of (piece ks, half h, sample s) inside a tensor block: the sample index is XOR-swizzled
__device__ __forceinline__ int dwb_unit(int ks, int h, int s) { return ks * 64 + h * 32 + (s ^ (4 * (2 * (ks & 1) + h))); }

// MFMA operand (A or B alike) of feature tile t, k-step kstep (samples 16 kstep .. +15) from a tensor block in LDS
__device__ __forceinline__ u32x4 dwb_operand(const unsigned char* blk, int t, int kstep, int lane) {
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, hh = lane >> 5;
  const int ks = 2 * t + (grp & 1), h = p & 1, e = p >> 1;
  const int s0 = 16 * kstep + 8 * hh + q;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(blk + dwb_unit(ks, h, s0) * 16 + e * 8));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(blk + dwb_unit(ks, h, s0 + 4) * 16 + e * 8));
  u32x4 r;
  r[0] = __builtin_bit_cast(uint2, lo).x;
  r[1] = __builtin_bit_cast(uint2, lo).y;
  r[2] = __builtin_bit_cast(uint2, hi).x;
  r[3] = __builtin_bit_cast(uint2, hi).y;
  return r;
}

template <int NIT>
__global__ __launch_bounds__(BF_WG, 1) void k_dw_bf16(const DwBfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int XKS = 2 * NIT;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  const int per = (a.wb_tot + gridDim.x - 1) / gridDim.x;
  const int b_lo = blockIdx.x * per, b_hi = min(a.wb_tot, b_lo + per);
  const int nb = b_hi - b_lo;
  const int gks = a.g_ks, total = gks + XKS;
  const bool worker = wv < a.o_tiles;

  f32x16 acc[NIT], accb;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NIT; ++i) acc[i] = zero;
  accb = zero;
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // bf16 1.0 pairs

  // pieces of a block are dealt round-robin to the waves; every wave issues the same NUMBER of loads per block (the
  // counted wait needs that): a wave whose turn falls beyond the last piece loads the last piece again
  constexpr int NPW = 4;  // ceil((16 + 16) / 8); also used for smaller tensors (duplicates are harmless)
  // swizzled source lane: LDS lane L of a piece holds sample (L & 31) ^ swz of half L >> 5
  auto dma_block = [&](int b, int slot) {
    const int wb = b_lo + (b < nb ? b : nb - 1);
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      int pc = wv + 8 * k;
      pc = pc < total ? pc : total - 1;
      const bool is_g = pc < gks;
      const int ks = is_g ? pc : pc - gks;
      const int h = lane >> 5, s = (lane & 31) ^ (4 * (2 * (ks & 1) + h));
      const unsigned char* src = is_g ? a.G + ((size_t)wb * gks + ks) * BF_FRAG_BYTES : a.X + ((size_t)wb * XKS + ks) * BF_FRAG_BYTES;
      glds16(src + (h * 32 + s) * 16, lds_base + slot * DWB_SLOT_BYTES + (is_g ? ks : 16 + ks) * BF_FRAG_BYTES);
    }
  };
  if (nb > 0) {
#pragma unroll
    for (int b = 0; b < DWB_NSLOT - 1; ++b) dma_block(b, b);
    for (int b = 0; b < nb; ++b) {
      wait_vmcnt<(DWB_NSLOT - 2) * NPW>();  // my pieces of block b are in LDS ...
      __builtin_amdgcn_s_barrier();         // ... and everybody's; everybody is done with block b - 1
      asm volatile("" ::: "memory");
      dma_block(b + DWB_NSLOT - 1, (b + DWB_NSLOT - 1) % DWB_NSLOT);  // into the slot of block b - 1
      if (worker) {
        const unsigned char* gb = lds + (b % DWB_NSLOT) * DWB_SLOT_BYTES;
        const unsigned char* xb = gb + 16 * BF_FRAG_BYTES;
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep) {
          const u32x4 A = dwb_operand(gb, wv, kstep, lane);
#pragma unroll
          for (int i = 0; i < NIT; ++i) acc[i] = bf_mfma(A, dwb_operand(xb, i, kstep, lane), acc[i]);
          accb = bf_mfma(A, ones, accb);
        }
      }
    }
    wait_vmcnt<0>();  // nothing may still be writing this workgroup's LDS when it ends
  }
  if (worker) {
    // accumulator layout: lane l holds column i = l & 31 and rows o = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    const int NI = NIT * 32, ld = NI + 1;
    float* slab = a.slabs + (size_t)blockIdx.x * (a.o_tiles * 32) * ld;
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int i = 0; i < NIT; ++i) slab[(size_t)o * ld + 32 * i + n] = acc[i][r];
      if (n == 0) slab[(size_t)o * ld + NI] = accb[r];
    }
  }
}

// dW[o_first + o][col0 + i] = sum over slabs, i < nin_real; db[o] likewise from the last slab column
struct DwBfReduceArgs {
  const float* slabs;
  int nslab, rows, ni;          // slab = [rows][ni + 1]
  int o_first, o_count, nin_real;
  float* dW; int ldw, col0;
  float* db;                    // or null
};

// 64 output elements per block; the slabs are split over 4 thread groups whose partial sums are added in a fixed order
__global__ __launch_bounds__(256) void k_dw_bf16_reduce(const DwBfReduceArgs a) {
  __shared__ float part[4][64];
  const int ld = a.ni + 1;
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  const bool in_range = idx < a.o_count * ld;
  const int o = in_range ? idx / ld : 0, i = in_range ? idx - o * ld : 0;
  const bool wanted = in_range && ((i < a.ni && i < a.nin_real) || (i == a.ni && a.db != nullptr));
  float s = 0.f;
  if (wanted) {
    const float* p = a.slabs + (size_t)(a.o_first + o) * ld + i;
    const size_t stride = (size_t)a.rows * ld;
    const int per = (a.nslab + 3) / 4, k0 = grp * per, k1 = min(a.nslab, k0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s0 += p[(size_t)k * stride];
      s1 += p[(size_t)(k + 1) * stride];
      s2 += p[(size_t)(k + 2) * stride];
      s3 += p[(size_t)(k + 3) * stride];
    }
    for (; k < k1; ++k) s0 += p[(size_t)k * stride];
    s = (s0 + s1) + (s2 + s3);
  }
  part[grp][e] = s;
  __syncthreads();
  if (grp == 0 && wanted) {
    const float t = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (i == a.ni)
      a.db[o] = t;
    else
      a.dW[(size_t)o * a.ldw + a.col0 + i] = t;
  }
}

size_t dw_bf16_slab_floats() { return (size_t)DWB_WGS * 256 * 257; }

// One weight-gradient GEMM.  G: g_ks pieces per wave block (o_tiles = ceil(g_ks / 2) row tiles), X: x_ks in {16, 8, 4, 2}.
hipError_t launch_dw_bf16(const unsigned char* G, int g_ks, const unsigned char* X, int x_ks, int wb_tot, float* slabs,
                          int o_first, int o_count, int nin_real, float* dW, int ldw, int col0, float* db, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dw_bf16<8>), hipFuncAttributeMaxDynamicSharedMemorySize, DWB_LDS_BYTES)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dw_bf16<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DWB_LDS_BYTES)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dw_bf16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DWB_LDS_BYTES)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dw_bf16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, DWB_LDS_BYTES)) != hipSuccess) return e;
    attr_done = true;
  }
  DwBfArgs a;
  a.G = G; a.X = X; a.g_ks = g_ks; a.o_tiles = (g_ks + 1) / 2; a.wb_tot = wb_tot; a.slabs = slabs;
  const int wgs = wb_tot < DWB_WGS ? wb_tot : DWB_WGS;
  switch (x_ks) {
    case 16: hipLaunchKernelGGL((k_dw_bf16<8>), dim3(wgs), dim3(BF_WG), DWB_LDS_BYTES, st, a); break;
    case 8: hipLaunchKernelGGL((k_dw_bf16<4>), dim3(wgs), dim3(BF_WG), DWB_LDS_BYTES, st, a); break;
    case 4: hipLaunchKernelGGL((k_dw_bf16<2>), dim3(wgs), dim3(BF_WG), DWB_LDS_BYTES, st, a); break;
    case 2: hipLaunchKernelGGL((k_dw_bf16<1>), dim3(wgs), dim3(BF_WG), DWB_LDS_BYTES, st, a); break;
    default: return hipErrorInvalidValue;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  DwBfReduceArgs r;
  r.slabs = slabs; r.nslab = wgs; r.rows = a.o_tiles * 32; r.ni = x_ks * 16;
  r.o_first = o_first; r.o_count = o_count; r.nin_real = nin_real; r.dW = dW; r.ldw = ldw; r.col0 = col0; r.db = db;
  const int n = o_count * (r.ni + 1);
  hipLaunchKernelGGL(k_dw_bf16_reduce, dim3((n + 63) / 64), dim3(256), 0, st, r);
  return hipGetLastError();
}

}  // namespace nerf
