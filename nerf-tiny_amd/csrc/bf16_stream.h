// bf16_stream.h -- the machinery shared by the bf16 field kernels (forward and backward chain): a workgroup of 8 waves
// consumes one stream of 1-KiB MFMA A fragments through an LDS ring filled by direct-to-LDS loads, each wave holding 32
// samples x all features in registers (see field_fwd_bf16.hip for the design notes).  MI355X / gfx950 only.
#pragma once
#include "bf16_common.h"
#include "field_common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3)))* lptr_t;

#ifndef NERF_BF_NS  // (timing experiments only: ring depth of the forward stream)
#define NERF_BF_NS 8
#endif
constexpr int BF_NS = NERF_BF_NS;  // LDS ring slots of the forward kernel (a stream description names its own: S::NS)
constexpr int BF_SYNC_POS = 8;  // fragment position inside a chunk at which the next chunk is published
constexpr int BF_D = 6;         // fragment reads in flight per wave (<= BF_CHUNK - BF_SYNC_POS); a stream names its own: S::D
constexpr int BF_EPI_POS = 2;   // k-step of the next tile at which a finished accumulator is consumed
constexpr int BF_WG = 512;      // 8 waves x 32 samples
constexpr int BF_LDS_BYTES = BF_BIAS_BYTES + BF_NS * BF_CHUNK * BF_FRAG_BYTES;
constexpr int BF_SMALL_MAX_WGS = 256;  // the training kernels run 4-wave workgroups (128 samples) up to this many of them: one per CU
// NERF_BF16_4WAVE=0: the training kernels keep 8-wave workgroups for small passes too (A/B measurements only)
inline bool bf16_four_waves_disabled() {
  static const bool off = [] { const char* e = getenv("NERF_BF16_4WAVE"); return e && atoi(e) == 0; }();
  return off;
}
static_assert(BF_D <= BF_CHUNK - BF_SYNC_POS, "a prefetched fragment must not lie in an unpublished chunk");

__device__ __forceinline__ unsigned pack2(float a, float b) {  // two fp32 -> two bf16 (RNE), a in the low half
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned pack2_relu(float a, float b) {
  const f32x2 v = {a, b};
  s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(v, bf16x2));
  const s16x2 z = {0, 0};
  s = __builtin_elementwise_max(s, z);  // bf16 as int16: negative values (sign bit) -> 0, positive order preserved
  return __builtin_bit_cast(unsigned, s);
}
// bit (15 - r) = accumulator register r is >= +0 ("alive": its ReLU passes the gradient); one v_alignbit per register
__device__ __forceinline__ unsigned alive_bits(const f32x16& A) {
  unsigned bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(A[r]), 31);
  return ~bits & 0xffffu;
}

// The lane id, produced AT this program point: what the epilogues of the stream kernels form their addresses from.  The zero the count starts
// from is an SGPR made opaque by an empty asm, so the v_mbcnt pair cannot be merged with an earlier copy (which would have to stay alive --
// or be parked in scratch -- across the stream), yet the pair itself is ordinary compiler-issued code: the hazard recognizer sees its
// destination and keeps it clear of the accumulator rows of an MFMA still in flight.  (Round 4's form was an `asm volatile` v_mbcnt whose
// destination the allocator once placed on a dead row of an in-flight MFMA's accumulator -- every lane of an epilogue then stored to sample
// 0 -- held off by 20 s_nop wait states the recognizer could not check.)
__device__ __forceinline__ unsigned lane_id_here() {
  unsigned z = 0u;
  asm volatile("" : "+s"(z));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>) -- every stream step has
// its own constants (LDS offsets, wait counts), so nothing is left to the loop unroller
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// DMA pieces per wave and chunk: 16 / (waves per workgroup).  The 8-wave streams say nothing (2); a 4-wave stream sets PW = 4.
template <class S, class = void> struct BfPW { static constexpr int v = 2; };
template <class S> struct BfPW<S, std::void_t<decltype(S::PW)>> { static constexpr int v = S::PW; };

struct BfCtx {
  const unsigned char* wimg;  // global: bias block + fragment stream
  unsigned char* lds;         // bias block + ring
  unsigned lds_base;          // the same as an LDS byte address
  int lane, wv;
};

// One direct-to-LDS load: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane*16 (lds_dst wave-uniform).
// Inline asm on purpose: hipcc treats a builtin LDS-DMA as a pending write to the whole LDS array and drains the load
// queue (vmcnt(0)) in front of unrelated ds_reads; hidden from it, the loads are ordered by bf_sync's counted waits alone.
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned lds_dst) {
#ifdef NERF_TIMING_NO_DMA  // (timing experiments only)
  return;
#endif
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// A 16-byte store of a saved piece (layer inputs, ReLU masks, pre-activation gradients: written once, read once by a later kernel,
// gigabytes per step) with the non-temporal hint: measured 3 % on the whole bf16 train step against plain stores (forward-with-saves
// 0.93 -> 0.87 ms, chains 0.96 -> 0.91, and the weight-gradient kernels that follow 1.45 -> 1.38: less dirty data parked in L2).
__device__ __forceinline__ void store_piece(unsigned char* dst, const u32x4& v) {
#if defined(NERF_SAVE_STORE_SC1)    // (timing experiment: write-through, line dropped from L2)
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#elif defined(NERF_SAVE_STORE_PLAIN)  // (timing experiment)
  *reinterpret_cast<u32x4*>(dst) = v;
#else
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
#endif
}

// this wave's two 1-KiB pieces of chunk c -> ring slot c % S::NS  (S::RING_OFF = LDS offset of the ring)
template <class S>
__device__ __forceinline__ void bf_dma_chunk(const BfCtx& c, int chunk) {
  const int slot = chunk % S::NS;
  constexpr int PW = BfPW<S>::v;
#pragma unroll
  for (int e = 0; e < PW; ++e) {
    const int fr = PW * c.wv + e;
    glds16(c.wimg + BF_BIAS_BYTES + ((size_t)chunk * BF_CHUNK + fr) * BF_FRAG_BYTES + c.lane * 16,
           c.lds_base + S::RING_OFF + (slot * BF_CHUNK + fr) * BF_FRAG_BYTES);
  }
}

// A stream description S provides: NFRAG, NCHUNK, NS, RING_OFF, PROLOGUE_STORES and stores_before(idx) = number of vector stores the
// wave has issued (in program order) before the sync point at stream step idx, not counting the prologue's.  Stores share
// the vmcnt queue with the weight loads and retire in order, so the counted wait must allow for the younger ones.
template <class S>
__device__ __forceinline__ constexpr int bf_wait_count(int c) {
  const int last = (c + S::NS - 2 < S::NCHUNK - 1) ? c + S::NS - 2 : S::NCHUNK - 1;  // newest chunk requested so far
  const int loads = (last >= c + 2) ? BfPW<S>::v * (last - (c + 2) + 1) : 0;         // younger than chunk c + 1's
  // chunk c + 1 was requested at the sync point of chunk c + 2 - NS (or in the prologue)
  const int c_req = c + 2 - S::NS;
  const int now = S::stores_before(c * BF_CHUNK + BF_SYNC_POS);
  const int then = c_req >= 0 ? S::stores_before(c_req * BF_CHUNK + BF_SYNC_POS) : -S::PROLOGUE_STORES;
  const int n = loads + (now - then);
  return n > 63 ? 63 : n;
}

// executed by every wave at fragment position BF_SYNC_POS of chunk c
template <class S, int chunk>
__device__ __forceinline__ void bf_sync(const BfCtx& c) {
#ifndef NERF_TIMING_NO_WAIT                   // (timing experiments only)
  wait_vmcnt<bf_wait_count<S>(chunk)>();     // my pieces of chunk + 1 are in LDS ...
#endif
#ifndef NERF_TIMING_NO_BARRIER               // (timing experiments only: results are wrong without it)
  __builtin_amdgcn_s_barrier();              // ... and so are everybody's; everybody is past chunk - 1
#endif
  asm volatile("" ::: "memory");             // no LDS read may be moved above the barrier by the compiler
  if (chunk + S::NS - 1 < S::NCHUNK) bf_dma_chunk<S>(c, chunk + S::NS - 1);  // into the slot of chunk - 1
}

// kernel prologue: (S::HAS_BIAS: bias block to LDS offset 0, 2 pieces per wave) and chunks 0 .. NS-2
template <class S>
__device__ __forceinline__ void bf_stream_start(const BfCtx& c) {
  if constexpr (S::HAS_BIAS) {
    constexpr int PW = BfPW<S>::v;
#pragma unroll
    for (int e = 0; e < PW; ++e) {
      const int fr = PW * c.wv + e;
      glds16(c.wimg + fr * BF_FRAG_BYTES + c.lane * 16, c.lds_base + fr * BF_FRAG_BYTES);
    }
  }
#pragma unroll
  for (int ch = 0; ch < S::NS - 1; ++ch) bf_dma_chunk<S>(c, ch);
}

template <class S>
__device__ __forceinline__ u32x4 bf_frag(const BfCtx& c, int idx) {
  const int slot = (idx / BF_CHUNK) % S::NS;
  return *reinterpret_cast<const u32x4*>(c.lds + S::RING_OFF + (slot * BF_CHUNK + idx % BF_CHUNK) * BF_FRAG_BYTES + c.lane * 16);
}

// everything requested before chunk 1 has landed (mine, then everybody's); first fragments into the register ring.
// The wave issued S::PROLOGUE_STORES stores after bf_stream_start.
template <class S>
__device__ __forceinline__ void bf_stream_first(const BfCtx& c, u32x4 (&fr)[S::D]) {
  constexpr int n = BfPW<S>::v * (S::NS - 2) + S::PROLOGUE_STORES;
  wait_vmcnt<(n > 63 ? 63 : n)>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_assert(S::D <= BF_CHUNK - BF_SYNC_POS, "a prefetched fragment must not lie in an unpublished chunk");
#pragma unroll
  for (int i = 0; i < S::D; ++i) fr[i] = bf_frag<S>(c, i);
}

__device__ __forceinline__ f32x16 bf_bias_tile(const BfCtx& c, int tile) {
  const int h = c.lane >> 5;
  f32x16 a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = *reinterpret_cast<const float4*>(c.lds + (tile * 32 + 8 * g + 4 * h) * 4);
    a[4 * g + 0] = q.x;
    a[4 * g + 1] = q.y;
    a[4 * g + 2] = q.z;
    a[4 * g + 3] = q.w;
  }
  return a;
}

__device__ __forceinline__ f32x16 bf_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One segment of the stream: NFT output tiles x (KSA + KSB) k-steps starting at fragment S0; inputs inA (k-steps
// 0..KSA-1) then inB.  fr = ring of BF_D prefetched fragments (fr[idx % BF_D] holds fragment idx on entry to step idx).
// Two accumulators alternate so that nothing waits for an MFMA result: tile f runs in acc[(P0 + f) & 1]; the finished
// accumulator of tile f-1 is consumed by epi(f-1, .) BF_EPI_POS k-steps into tile f (the last tile of the previous
// segment by prev_epi), and right after that it is re-started at the bias of tile f+1 (or of the next segment's tile 0,
// bias tile NEXT_BT; BT0 < 0: no biases, accumulators start at zero), an LDS read with >= 1 k-steps of MFMAs to land in.
template <class S, int S0, int NFT, int KSA, int KSB, int BT0, int P0, int NEXT_BT, class Epi, class PrevEpi>
__device__ __forceinline__ void bf_segment(const BfCtx& c, u32x4 (&fr)[S::D], f32x16 (&acc)[2], const u32x4* inA, const u32x4* inB,
                                           Epi&& epi, PrevEpi&& prev_epi) {
  constexpr int KS = KSA + KSB;
  static_assert(KS > BF_EPI_POS + 1, "segment too short for the deferred epilogue");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  static_for<NFT * KS>([&](auto I) {
    constexpr int f = I / KS, ks = I % KS, idx = S0 + I;
    constexpr int cur = (P0 + f) & 1, oth = (P0 + f + 1) & 1;
    if constexpr (idx % BF_CHUNK == BF_SYNC_POS) bf_sync<S, idx / BF_CHUNK>(c);
    const u32x4 a = fr[idx % S::D];
#ifndef NERF_TIMING_NO_FRAG  // (timing experiments only)
    if constexpr (idx + S::D < S::NFRAG) fr[idx % S::D] = bf_frag<S>(c, idx + S::D);
#endif
    if constexpr (ks < KSA)
      acc[cur] = bf_mfma(a, inA[ks], acc[cur]);
    else
      acc[cur] = bf_mfma(a, inB[ks - KSA], acc[cur]);
    if constexpr (ks == BF_EPI_POS) {
      if constexpr (f == 0)
        prev_epi(acc[oth]);
      else
        epi(f - 1, acc[oth]);
      if constexpr (f + 1 < NFT)
        acc[oth] = BT0 >= 0 ? bf_bias_tile(c, BT0 + f + 1) : zero;
      else if constexpr (NEXT_BT >= 0)
        acc[oth] = bf_bias_tile(c, NEXT_BT);
      else
        acc[oth] = zero;
    }
  });
}

// ---- a stream's store schedule: tiles in stream order, each with the number of stores its epilogue issues -------------
// The epilogue of the tile that ENDS at fragment index `end` (exclusive) runs in step end + BF_EPI_POS, after that step's
// sync point; so it precedes the sync point of step idx iff end + BF_EPI_POS < idx.
template <int NFRAG_>
struct BfStoreTable {
  int cum[NFRAG_ + 1];  // cum[i] = stores of all tiles with end + BF_EPI_POS < i
};

}  // namespace nerf
