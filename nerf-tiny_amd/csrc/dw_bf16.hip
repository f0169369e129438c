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

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace nerf {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_p;

constexpr int DWB_WGS = 256;

struct DwBfArgs {
  // ngemm same-shaped products share one launch: workgroup b works on product b % ngemm with the samples split over the
  // gridDim.x / ngemm workgroups of that product -- all CUs busy with 1/ngemm of the slabs per product
  const unsigned char* G[6];   // start of the gradient tensor (fragment layout), g_ks pieces per wave block
  const unsigned char* X1[6];  // input tensor(s): the i index runs over X1's x1_ks pieces, then X2's (2 * NIT - x1_ks)
  const unsigned char* X2;     // (single products only)
  int ngemm;
  const unsigned char* Z;   // HAS_Z: a second, 2-piece gradient tensor whose product with the X tiles z_tile0 .. z_tile0 + 7 is formed as well (X^T Z)
  int z_tile0;
  int g_ks, o_tiles;        // o_tiles = output row tiles (waves w >= o_tiles only help loading)
  int x1_ks;
  int wb_tot;
  float* slabs;             // [product][workgroup][o_tiles*32 (+32 with Z)][NIT*32 + 1]  (last column: sum of G over the samples)
  // split-fp32 train step (SPLIT instantiations): every operand is a (hi, mid) pair of bf16 tensors of the same layout; the mid part of a
  // gradient-type tensor (G, Z) lies gdelta bytes behind its hi part, the mid part of an input tensor (X1, X2) xdelta bytes
  long long gdelta, xdelta;
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

template <int NIT, bool HAS_Z, bool SPLIT = false>
struct DwbGeom {
  static constexpr int XKS = 2 * NIT;
  static constexpr int PIECES = 16 + XKS + (HAS_Z ? 2 : 0);   // slot layout: G at piece 0, X at 16, Z at 16 + XKS; SPLIT: the mid parts behind, same order
  static constexpr int SLOT_PIECES = (SPLIT ? 2 : 1) * PIECES;
  static constexpr int SLOT_BYTES = SLOT_PIECES * BF_FRAG_BYTES;
  // ring slots: three blocks in flight are enough where a block is 32+ KiB; the products with small blocks (layer 0: 20 KiB, colour head:
  // 10 KiB used) were paced by the per-block latency (3.4-3.8 TB/s, profiles/r03_train_bf16_pmc.json): they get as many slots as fit.
  // SPLIT (two-part operands: twice the bytes per block): what fits 160 KiB, at most four -- two for the 256 x 256 products (64 KiB per block:
  // one being multiplied, one in flight)
  static constexpr int NSLOT_FIT = (160 * 1024) / SLOT_BYTES;
  static constexpr int NSLOT = SPLIT ? (NSLOT_FIT > 4 ? 4 : NSLOT_FIT) : (PIECES <= 20 ? 7 : (PIECES <= 26 ? 6 : 4));
  static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES;
  static constexpr int NPW = (SLOT_PIECES + 7) / 8;           // loads per wave and block
  static_assert(NSLOT >= 2 && LDS_BYTES <= 160 * 1024, "LDS");
};

// The body of one workgroup: product `gi` of the argument block, workgroup `wg` of the `nwg` that share that product's samples; its
// partial sums go to slab_base + wg * rows * ld.
template <int NIT, bool HAS_Z, bool SPLIT = false>
__device__ __forceinline__ void dw_bf16_body(const DwBfArgs& a, unsigned char* lds, const int gi, const int wg, const int nwg, float* slab_base) {
  using Geo = DwbGeom<NIT, HAS_Z, SPLIT>;
  constexpr int XKS = Geo::XKS, NSLOT = Geo::NSLOT, NPW = Geo::NPW;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
  const unsigned char* const gG = a.G[gi];
  const unsigned char* const gX1 = a.X1[gi];
  const int per = (a.wb_tot + nwg - 1) / nwg;
  const int b_lo = wg * per, b_hi = min(a.wb_tot, b_lo + per);
  const int nb = b_hi - b_lo;
  const int gks = a.g_ks, x1 = a.x1_ks, total = gks + XKS + (HAS_Z ? 2 : 0);
  const bool worker = wv < a.o_tiles, zworker = HAS_Z && a.z_tile0 + wv < NIT;
  const int zt = a.z_tile0 + wv;  // the X tile this wave multiplies with Z

  f32x16 acc[NIT], accb, accz;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NIT; ++i) acc[i] = zero;
  accb = zero;
  accz = zero;
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // bf16 1.0 pairs

  // pieces of a block are dealt round-robin to the waves; every wave issues the same NUMBER of loads per block (the
  // counted wait needs that): a wave whose turn falls beyond the last piece loads the last piece again.
  // Swizzle on the source side: LDS lane L of a piece holds sample (L & 31) ^ swz of half L >> 5.
  auto dma_block = [&](int b, int slot) {
    const int wb = b_lo + (b < nb ? b : nb - 1);
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      int pc = wv + 8 * k;
      constexpr int SETS = SPLIT ? 2 : 1;
      pc = pc < SETS * total ? pc : SETS * total - 1;
      const bool mid = SPLIT && pc >= total;  // SPLIT: pieces [total, 2 total) are the mid parts, slot pieces Geo::PIECES ..
      if (mid) pc -= total;
      const long long gd = mid ? a.gdelta : 0, xd = mid ? a.xdelta : 0;
      const unsigned char* src;
      int ks, dst;
      // (timing experiments only, results wrong: NERF_TIMING_DW_HALF_G / _X read every second piece twice -> half the distinct bytes of
      // that operand reach HBM, the instruction stream and the LDS image stay as they are: the upper bound of what a design that moves
      // half the bytes -- alternate-layer recompute for X, zero-compaction for both -- could gain in this phase)
#ifdef NERF_TIMING_DW_HALF_G
#define DWB_GSRC(k) ((k) >> 1)
#else
#define DWB_GSRC(k) (k)
#endif
#ifdef NERF_TIMING_DW_HALF_X
#define DWB_XSRC(k) ((k) >> 1)
#else
#define DWB_XSRC(k) (k)
#endif
      if (pc < gks) {
        ks = pc; dst = ks;
        src = gG + gd + ((size_t)wb * gks + DWB_GSRC(ks)) * BF_FRAG_BYTES;
      } else if (pc < gks + XKS) {
        const int x = pc - gks;
        ks = x; dst = 16 + x;  // parity of the slot piece = parity of x (x1_ks is even)
        src = (x < x1 ? gX1 + ((size_t)wb * x1 + DWB_XSRC(x)) * BF_FRAG_BYTES : a.X2 + ((size_t)wb * (XKS - x1) + DWB_XSRC(x - x1)) * BF_FRAG_BYTES) + xd;
      } else {
        ks = pc - gks - XKS; dst = 16 + XKS + ks;
        src = a.Z + gd + ((size_t)wb * 2 + ks) * BF_FRAG_BYTES;
      }
      if (mid) dst += Geo::PIECES;
      const int h = lane >> 5, s = (lane & 31) ^ (4 * (2 * (ks & 1) + h));
      glds16(src + (h * 32 + s) * 16, lds_base + slot * Geo::SLOT_BYTES + dst * BF_FRAG_BYTES);  // (the non-temporal form measured the same)
    }
  };
  if (nb > 0) {
#pragma unroll
    for (int b = 0; b < NSLOT - 1; ++b) dma_block(b, b);
    for (int b = 0; b < nb; ++b) {
      wait_vmcnt<(NSLOT - 2) * NPW>();  // my pieces of block b are in LDS ...
      __builtin_amdgcn_s_barrier();     // ... and everybody's; everybody is done with block b - 1
      asm volatile("" ::: "memory");
      dma_block(b + NSLOT - 1, (b + NSLOT - 1) % NSLOT);  // into the slot of block b - 1
      const unsigned char* gb = lds + (b % NSLOT) * Geo::SLOT_BYTES;
      const unsigned char* xb = gb + 16 * BF_FRAG_BYTES;
      constexpr int MID = Geo::PIECES * BF_FRAG_BYTES;  // SPLIT: the mid parts of the block sit this far behind their hi parts
      if (worker) {
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep) {
          const u32x4 A = dwb_operand(gb, wv, kstep, lane);
          if constexpr (SPLIT) {  // G^T X = G_hi^T X_hi + G_hi^T X_mid + G_mid^T X_hi (+ O(2^-16)), the small products first; column sums of G_hi + G_mid
            const u32x4 Am = dwb_operand(gb + MID, wv, kstep, lane);
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
              const u32x4 Bh = dwb_operand(xb, i, kstep, lane);
              acc[i] = bf_mfma(Am, Bh, acc[i]);
              acc[i] = bf_mfma(A, dwb_operand(xb + MID, i, kstep, lane), acc[i]);
              acc[i] = bf_mfma(A, Bh, acc[i]);
            }
            accb = bf_mfma(Am, ones, accb);
            accb = bf_mfma(A, ones, accb);
          } else {
#pragma unroll
            for (int i = 0; i < NIT; ++i) acc[i] = bf_mfma(A, dwb_operand(xb, i, kstep, lane), acc[i]);
            accb = bf_mfma(A, ones, accb);
          }
        }
      }
      if (zworker) {  // rows = this wave's X tile, columns = Z features: (X^T Z) tile
        const unsigned char* zb = xb + XKS * BF_FRAG_BYTES;
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep) {
          const u32x4 Xh = dwb_operand(xb, zt, kstep, lane), Zh = dwb_operand(zb, 0, kstep, lane);
          if constexpr (SPLIT) {
            accz = bf_mfma(dwb_operand(xb + MID, zt, kstep, lane), Zh, accz);
            accz = bf_mfma(Xh, dwb_operand(zb + MID, 0, kstep, lane), accz);
          }
          accz = bf_mfma(Xh, Zh, accz);
        }
      }
    }
    wait_vmcnt<0>();  // nothing may still be writing this workgroup's LDS when it ends
  }
  // accumulator layout: lane l holds column l & 31 and rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
  const int NI = NIT * 32, ld = NI + 1;
  const int rows = a.o_tiles * 32 + (HAS_Z ? 32 : 0);
  float* slab = slab_base + (size_t)wg * rows * ld;
  const int n = lane & 31, hh = lane >> 5;
  if (worker) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int i = 0; i < NIT; ++i) slab[(size_t)o * ld + 32 * i + n] = acc[i][r];
      if (n == 0) slab[(size_t)o * ld + NI] = accb[r];
    }
  }
  if (zworker) {  // slab row o_tiles*32 + z, column i
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * zt + (r & 3) + 8 * (r >> 2) + 4 * hh;
      slab[(size_t)(a.o_tiles * 32 + n) * ld + i] = accz[r];
    }
  }
}

template <int NIT, bool HAS_Z, bool SPLIT = false>
__global__ __launch_bounds__(BF_WG, 1) void k_dw_bf16(const DwBfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int gi = blockIdx.x % a.ngemm, wg = blockIdx.x / a.ngemm, nwg = gridDim.x / a.ngemm;
  const int rows = a.o_tiles * 32 + (HAS_Z ? 32 : 0);
  dw_bf16_body<NIT, HAS_Z, SPLIT>(a, lds, gi, wg, nwg, a.slabs + (size_t)gi * nwg * rows * (NIT * 32 + 1));
}

// ---- SMALL batches: every product of the step (or of its early / late part) in ONE launch -------------------------------------------
// A launch per product hands each of them ALL workgroups and so a slab per workgroup: 219 MB of partial sums written and read back per
// step whatever the batch -- a third of the phase at 512 rays (0.31 ms against 0.17 for an eighth of the 4096-ray phase) -- behind five
// launch boundaries and with 12-block streams whose ring never fills.  Here the workgroups of ONE launch are dealt out over the
// products in proportion to their bytes per wave block: as many CUs busy, streams of 100+ blocks, a quarter of the slabs.  Products of
// different shapes run different instantiations of the same body behind a block-uniform switch.
struct DwBfMulti {
  static constexpr int MAXP = 12;
  DwBfArgs a[MAXP];     // per product (ngemm = 1, product index 0); a[i].slabs = its slab base
  int kind[MAXP];       // instantiation: 2 * NIT + HAS_Z
  int wg0[MAXP + 1];    // first workgroup of product i (wg0[n] = grid size)
  int n;
};

template <bool SPLIT>
__global__ __launch_bounds__(BF_WG, 1) void k_dw_bf16_multi(const DwBfMulti m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int i = 0;
#pragma unroll 1
  while (i + 1 < m.n && (int)blockIdx.x >= m.wg0[i + 1]) ++i;
  i = __builtin_amdgcn_readfirstlane(i);
  const DwBfArgs& a = m.a[i];
  const int wg = blockIdx.x - m.wg0[i], nwg = m.wg0[i + 1] - m.wg0[i];
  switch (m.kind[i]) {
    case 2 * 10 + 0: dw_bf16_body<10, false, SPLIT>(a, lds, 0, wg, nwg, a.slabs); break;
    case 2 * 9 + 1: dw_bf16_body<9, true, SPLIT>(a, lds, 0, wg, nwg, a.slabs); break;
    case 2 * 8 + 0: dw_bf16_body<8, false, SPLIT>(a, lds, 0, wg, nwg, a.slabs); break;
    case 2 * 4 + 0: dw_bf16_body<4, false, SPLIT>(a, lds, 0, wg, nwg, a.slabs); break;
    case 2 * 2 + 0: dw_bf16_body<2, false, SPLIT>(a, lds, 0, wg, nwg, a.slabs); break;
    default: break;
  }
}

// 64 output elements per block; the slabs are split over 4 thread groups whose partial sums are added in a fixed order
__global__ __launch_bounds__(256) void k_dw_bf16_reduce(const DwBfReduceArgs a) {
  __shared__ float part[4][64];
  const int ld = a.ni + 1;
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  const bool in_range = idx < a.o_count * ld;
  const int o = in_range ? idx / ld : 0, i = in_range ? idx - o * ld : 0;
  const bool wanted = in_range && ((i < a.ni && i >= a.i_first && i < a.i_first + a.i_count && a.dW != nullptr) || (i == a.ni && a.db != nullptr));
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
      a.dW[(size_t)o * a.ldw + a.col0 + (i - a.i_first)] = t;
  }
}

// ONE launch for all slab sums of a step: blockIdx.y = descriptor (the thirteen per-product launches cost 90 us per step in launch
// gaps and tiny grids -- a quarter of the weight-gradient phase of a 512-ray step)
__global__ __launch_bounds__(256) void k_dw_bf16_reduce_batch(const DwBfReduceBatch b) {
  __shared__ float part[4][64];
  const DwBfReduceArgs& a = b.r[blockIdx.y];
  const int ld = a.ni + 1;
  if ((int)blockIdx.x * 64 >= a.o_count * ld) return;  // (block-uniform)
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  const bool in_range = idx < a.o_count * ld;
  const int o = in_range ? idx / ld : 0, i = in_range ? idx - o * ld : 0;
  const bool wanted = in_range && ((i < a.ni && i >= a.i_first && i < a.i_first + a.i_count && a.dW != nullptr) || (i == a.ni && a.db != nullptr));
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
      a.dW[(size_t)o * a.ldw + a.col0 + (i - a.i_first)] = t;
  }
}

hipError_t launch_dw_bf16_reduce_batch(const DwBfReduceBatch& b, hipStream_t st) {
  int most = 1;
  for (int k = 0; k < b.n; ++k) {
    const int blocks = (b.r[k].o_count * (b.r[k].ni + 1) + 63) / 64;
    most = blocks > most ? blocks : most;
  }
  hipLaunchKernelGGL(k_dw_bf16_reduce_batch, dim3(most, b.n), dim3(256), 0, st, b);
  return hipGetLastError();
}

// every product of a step keeps its own slabs (ONE reduce launch at the end): layer 0, the six grouped 256 x 256 products, layer 4, the
// folded dpre_dir product with the sigma rows, the colour head
size_t dw_bf16_slab_floats() {
  return (size_t)DWB_WGS * 256 * 65 + (size_t)6 * (DWB_WGS / 6) * 256 * 257 + (size_t)DWB_WGS * 256 * 321 + (size_t)DWB_WGS * 160 * 289 + (size_t)DWB_WGS * 32 * 129;
}

template <int NIT, bool HAS_Z, bool SPLIT>
static hipError_t dwb_launch_as(const DwBfArgs& a, int wgs, hipStream_t st) {
  static std::atomic<unsigned long long> opted{0};
  using Geo = DwbGeom<NIT, HAS_Z, SPLIT>;
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_dw_bf16<NIT, HAS_Z, SPLIT>)}, Geo::LDS_BYTES)) return e;
  hipLaunchKernelGGL((k_dw_bf16<NIT, HAS_Z, SPLIT>), dim3(wgs), dim3(BF_WG), Geo::LDS_BYTES, st, a);
  return hipGetLastError();
}
// (a.gdelta / a.xdelta != 0: the two-part form of the split-fp32 train step)
template <int NIT, bool HAS_Z>
static hipError_t dwb_launch(const DwBfArgs& a, int wgs, hipStream_t st) {
  return (a.gdelta || a.xdelta) ? dwb_launch_as<NIT, HAS_Z, true>(a, wgs, st) : dwb_launch_as<NIT, HAS_Z, false>(a, wgs, st);
}

// One pass over the samples: slabs <- G^T [X1 | X2]  (+ [X1 | X2]^T Z in 32 extra slab rows).  x1_ks + x2_ks in {2,4,8,16,18,20}.
hipError_t launch_dw_bf16_gemm(const unsigned char* G, int g_ks, const unsigned char* X1, int x1_ks, const unsigned char* X2, int x2_ks,
                               const unsigned char* Z, int wb_tot, float* slabs, int* nslab, hipStream_t st, DwBfSplit sp) {
  DwBfArgs a;
  memset(&a, 0, sizeof(a));
  a.gdelta = sp.gdelta; a.xdelta = sp.xdelta;
  a.z_tile0 = Z ? x1_ks / 2 : 0;  // Z multiplies the X2 tiles (the sigma head rides on the product whose second input tensor is h7)
  a.G[0] = G; a.X1[0] = X1; a.ngemm = 1; a.X2 = X2 ? X2 : X1; a.Z = Z; a.g_ks = g_ks; a.o_tiles = (g_ks + 1) / 2; a.x1_ks = x1_ks; a.wb_tot = wb_tot; a.slabs = slabs;
  const int wgs = wb_tot < DWB_WGS ? wb_tot : DWB_WGS;
  *nslab = wgs;
  const int xks = x1_ks + x2_ks;
  if (Z) return xks == 18 ? dwb_launch<9, true>(a, wgs, st) : xks == 16 ? dwb_launch<8, true>(a, wgs, st) : hipErrorInvalidValue;
  switch (xks) {
    case 20: return dwb_launch<10, false>(a, wgs, st);
    case 18: return dwb_launch<9, false>(a, wgs, st);
    case 16: return dwb_launch<8, false>(a, wgs, st);
    case 8: return dwb_launch<4, false>(a, wgs, st);
    case 4: return dwb_launch<2, false>(a, wgs, st);
    case 2: return dwb_launch<1, false>(a, wgs, st);
    default: return hipErrorInvalidValue;
  }
}

// n (<= 6) products G_k^T X_k of 256 x 256 outputs in one launch; product k's slabs start at slabs + k * (*nslab) * 256 * 257
hipError_t launch_dw_bf16_group(const unsigned char* const* Gs, const unsigned char* const* Xs, int n, int wb_tot, float* slabs, int* nslab,
                                hipStream_t st, DwBfSplit sp) {
  if (n < 1 || n > 6) return hipErrorInvalidValue;
  DwBfArgs a;
  memset(&a, 0, sizeof(a));
  a.gdelta = sp.gdelta; a.xdelta = sp.xdelta;
  for (int k = 0; k < n; ++k) { a.G[k] = Gs[k]; a.X1[k] = Xs[k]; }
  a.X2 = Xs[0]; a.ngemm = n; a.g_ks = 16; a.o_tiles = 8; a.x1_ks = 16; a.wb_tot = wb_tot; a.slabs = slabs;
  int nwg = DWB_WGS / n;
  if (nwg > wb_tot) nwg = wb_tot;
  *nslab = nwg;
  return dwb_launch<8, false>(a, nwg * n, st);
}

// n (<= 12) products of ANY of the shapes above in ONE launch of (at most) DWB_WGS workgroups, dealt out in proportion to the bytes a
// product reads per wave block (largest remainder, at least one each, never more than wave blocks).  Fills p[i].slabs / p[i].nslab
// (carved from slab_base in order) and returns the end of the used slab space in *slab_end.
hipError_t launch_dw_bf16_multi(DwBfProd* p, int n, int wb_tot, float* slab_base, const float* slab_limit, float** slab_end, hipStream_t st, DwBfSplit sp) {
  if (n < 1 || n > DwBfMulti::MAXP || wb_tot < 1) return hipErrorInvalidValue;
  DwBfMulti m;
  memset(&m, 0, sizeof(m));
  m.n = n;
  // cost of a product per wave block = its KiB, weighted by how far below the big products' rate its shape streams (small blocks are paced
  // by the ring's per-block latency: profiles/r03_train_bf16_pmc.json -- 256 x 256 products 5.8 TB/s, layer 4 5.4, the folded product 4.3,
  // colour head 3.8, layer 0 3.4); NERF_DW_BF16_COST=0: plain bytes (A/B measurements only)
  static const bool by_cost = [] { const char* e = getenv("NERF_DW_BF16_COST"); return !(e && atoi(e) == 0); }();
  // NERF_DW_BF16_COSTS="l0,col,fold,l4" (per cent; tuning sweeps only) replaces the shape factors below
  static const struct Costs { int l0, col, fold, l4; } costs = [] {
    Costs c{170, 150, 135, 107};
    if (const char* e = getenv("NERF_DW_BF16_COSTS")) sscanf(e, "%d,%d,%d,%d", &c.l0, &c.col, &c.fold, &c.l4);
    return c;
  }();
  int pieces[DwBfMulti::MAXP], total = 0;
  for (int i = 0; i < n; ++i) {
    const int kib = p[i].g_ks + p[i].x1_ks + p[i].x2_ks + (p[i].Z ? 2 : 0), xks = p[i].x1_ks + p[i].x2_ks;
    const int pct = !by_cost ? 100 : xks == 4 ? costs.l0 : xks == 8 ? costs.col : (xks == 18 && p[i].Z) ? costs.fold : xks == 20 ? costs.l4 : 100;
    pieces[i] = kib * pct;
    total += pieces[i];
  }
  const int budget = DWB_WGS;
  int nwg[DwBfMulti::MAXP], used = 0;
  for (int i = 0; i < n; ++i) {
    nwg[i] = (int)((long long)budget * pieces[i] / total);
    if (nwg[i] < 1) nwg[i] = 1;
    if (nwg[i] > wb_tot) nwg[i] = wb_tot;
    used += nwg[i];
  }
  // hand the remaining workgroups to the products with the most bytes per workgroup (fixed order: deterministic slabs)
  while (used < budget) {
    int best = -1;
    double worst = 0.0;
    for (int i = 0; i < n; ++i) {
      const double load = (double)pieces[i] / nwg[i];
      if (nwg[i] < wb_tot && load > worst) { worst = load; best = i; }
    }
    if (best < 0) break;
    ++nwg[best]; ++used;
  }
  while (used > budget) {  // (only when n > budget or the minimum of one each overshoots: not with 12 products)
    int best = -1;
    double least = 1e30;
    for (int i = 0; i < n; ++i) {
      const double load = (double)pieces[i] / nwg[i];
      if (nwg[i] > 1 && load < least) { least = load; best = i; }
    }
    if (best < 0) break;
    --nwg[best]; --used;
  }
  float* slab = slab_base;
  int wg0 = 0;
  for (int i = 0; i < n; ++i) {
    DwBfArgs& a = m.a[i];
    const int xks = p[i].x1_ks + p[i].x2_ks, nit = xks / 2;
    const bool z = p[i].Z != nullptr;
    if (xks % 2 || !((nit == 10 && !z) || (nit == 9 && z) || (nit == 8 && !z) || (nit == 4 && !z) || (nit == 2 && !z))) return hipErrorInvalidValue;
    a.G[0] = p[i].G; a.X1[0] = p[i].X1; a.X2 = p[i].X2 ? p[i].X2 : p[i].X1; a.ngemm = 1; a.Z = p[i].Z; a.z_tile0 = z ? p[i].x1_ks / 2 : 0;
    a.g_ks = p[i].g_ks; a.o_tiles = (p[i].g_ks + 1) / 2; a.x1_ks = p[i].x1_ks; a.wb_tot = wb_tot; a.slabs = slab;
    a.gdelta = sp.gdelta; a.xdelta = sp.xdelta;
    m.kind[i] = 2 * nit + (z ? 1 : 0);
    m.wg0[i] = wg0;
    wg0 += nwg[i];
    p[i].slabs = slab;
    p[i].nslab = nwg[i];
    const size_t rows = (size_t)a.o_tiles * 32 + (z ? 32 : 0);
    slab += (size_t)nwg[i] * rows * (nit * 32 + 1);
  }
  m.wg0[n] = wg0;
  if (slab_end) *slab_end = slab;
  if (slab_limit && slab > slab_limit) return hipErrorOutOfMemory;  // the slabs of this plan would run past the workspace's slab space: nothing is enqueued
  static std::atomic<unsigned long long> opted{0}, opted_split{0};
  constexpr int lds_bytes = 144 * 1024;
  static_assert(DwbGeom<10, false>::LDS_BYTES <= lds_bytes && DwbGeom<9, true>::LDS_BYTES <= lds_bytes && DwbGeom<8, false>::LDS_BYTES <= lds_bytes &&
                DwbGeom<4, false>::LDS_BYTES <= lds_bytes && DwbGeom<2, false>::LDS_BYTES <= lds_bytes, "LDS of the multi-product launch");
  static_assert(DwbGeom<10, false, true>::LDS_BYTES <= lds_bytes && DwbGeom<9, true, true>::LDS_BYTES <= lds_bytes && DwbGeom<8, false, true>::LDS_BYTES <= lds_bytes &&
                DwbGeom<4, false, true>::LDS_BYTES <= 160 * 1024 && DwbGeom<2, false, true>::LDS_BYTES <= 160 * 1024, "LDS of the two-part multi-product launch");
  if (sp.gdelta || sp.xdelta) {
    constexpr int lds_split = 160 * 1024;
    if (hipError_t e = ensure_dynamic_lds(opted_split, {reinterpret_cast<const void*>(&k_dw_bf16_multi<true>)}, lds_split)) return e;
    hipLaunchKernelGGL(k_dw_bf16_multi<true>, dim3(wg0), dim3(BF_WG), lds_split, st, m);
    return hipGetLastError();
  }
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(&k_dw_bf16_multi<false>)}, lds_bytes)) return e;
  hipLaunchKernelGGL(k_dw_bf16_multi<false>, dim3(wg0), dim3(BF_WG), lds_bytes, st, m);
  return hipGetLastError();
}

// slab rows [o_first, o_first + o_count), columns [i_first, i_first + i_count) -> dW[o][col0 + i]; last slab column -> db
hipError_t launch_dw_bf16_reduce(const float* slabs, int nslab, int rows, int ni, int o_first, int o_count, int i_first, int i_count,
                                 float* dW, int ldw, int col0, float* db, hipStream_t st) {
  DwBfReduceArgs r;
  r.slabs = slabs; r.nslab = nslab; r.rows = rows; r.ni = ni;
  r.o_first = o_first; r.o_count = o_count; r.i_first = i_first; r.i_count = i_count; r.dW = dW; r.ldw = ldw; r.col0 = col0; r.db = db;
  const int n = o_count * (ni + 1);
  hipLaunchKernelGGL(k_dw_bf16_reduce, dim3((n + 63) / 64), dim3(256), 0, st, r);
  return hipGetLastError();
}

}  // namespace nerf
