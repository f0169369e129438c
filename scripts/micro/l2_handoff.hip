// Microbenchmark for DESIGN.md section 9's "layer-stationary pipeline" question (round-3 review, item 3b): can activation tiles be
// handed from CU to CU through memory fast enough that a CU group per layer (weights of one 256 x 256 bf16 layer resident in LDS)
// would beat today's materialise-in-HBM structure of the bf16 train step?
//
// What it runs.  Workgroups of 8 waves, one per CU (144 KiB of LDS), placed by block index: block b sits on XCD b % 8 (round-robin
// dispatch), local index k = b / 8; consecutive local indices of one XCD form a CHAIN of `stages` CUs.  Every stage holds the 128 KiB
// A-fragment image of one 256 x 256 bf16 layer in LDS and, per 256-sample tile (8 waves x 32 samples), runs the layer exactly as
// field_fwd_bf16.hip does (128 v_mfma_f32_32x32x16_bf16 per wave and tile, fragments by ds_read_b128, the ReLU'd accumulators packed
// pairwise to bf16 = the next layer's B operands).  Stage 0 makes its inputs up; stage s > 0 takes them from stage s - 1 through a ring of
// R slots of 128 KiB in device memory (16 pieces of 1 KiB per wave and tile = the fragment layout of the save buffers):
//   mode 0  no hand-off at all (every stage makes its inputs up, nothing stored): the MFMA rate of the layer itself = the yardstick
//   mode 1  sc1 (write-through) 16-byte stores, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane publishes the
//           tile number in a flag word (agent-scope relaxed atomic = sc1 store); the consumer polls that word with ONE lane (sc1 load),
//           barrier, every wave loads its 16 pieces with sc1 16-byte loads (no acquire: MI355X_MICROARCH.md "visibility", table row 1);
//           the consumer acknowledges a slot with a second flag word so that the producer may overwrite it (ring back-pressure)
//   mode 2  plain stores + agent-scope release fence by one lane + flag; consumer: poll, agent-scope acquire fence, barrier, plain loads
// Reported per mode: ns per tile and stage, the hand-off rate per chain / XCD / chip (128 KiB x tiles / time), and MFMA-busy of a stage
// relative to mode 0.  `only_xcd` >= 0 keeps the chains of ONE XCD and retires every other block at once (the review's "on ONE XCD").
// Every spin is bounded (a fault sets a timeout word and the run is reported invalid; nothing can hang).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_handoff.hip -o scripts/micro/bin/l2_handoff && scripts/micro/bin/l2_handoff
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

constexpr int WG = 512, TILE_BYTES = 128 * 1024, W_BYTES = 128 * 1024, PIECE = 1024;
constexpr unsigned SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ u32x4 rnd(unsigned seed) {  // 8 bf16 values in (-1, 1), small enough that 8 layers do not overflow
  u32x4 r;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = hash(seed * 4 + i);
    const unsigned lo = 0x3d80u | (h & 0x7fu) | ((h >> 7) & 1u) << 15, hi = 0x3d80u | ((h >> 8) & 0x7fu) | ((h >> 15) & 1u) << 15;
    r[i] = lo | hi << 16;
  }
  return r;
}
__device__ __forceinline__ unsigned pack2_relu(float a, float b) {
  const f32x2 v = {a, b};
  s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(v, bf16x2));
  const s16x2 z = {0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(unsigned, s);
}

struct Args {
  unsigned char* ring;     // [chain][stage - 1][R][TILE_BYTES]
  unsigned* ready;         // [chain][stage - 1]: tiles published by the producer of this link (64-byte spaced)
  unsigned* acked;         // [chain][stage - 1]: tiles consumed by the consumer of this link
  unsigned* timeout;       // one word: set by any bounded spin that gave up
  float* sink;             // one float per thread (keeps the work alive)
  unsigned long long* cycles;  // [block]: cycles of the tile loop
  int stages, R, tiles, mode, only_xcd, chains_per_xcd;
};

__device__ __forceinline__ bool spin_until_ge(unsigned* word, unsigned want, unsigned* timeout) {
  unsigned n = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if (++n >= SPIN_LIMIT) { atomicOr(timeout, 1u); return false; }
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__global__ __launch_bounds__(WG, 1) void k_chain(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
  if (a.only_xcd >= 0 && xcd != a.only_xcd) return;
  const int chain_local = k / a.stages, stage = k % a.stages;
  if (chain_local >= a.chains_per_xcd) return;
  const int chain = xcd * a.chains_per_xcd + chain_local;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // the layer's A fragments: 8 tiles x 16 k-steps x 1 KiB, made up once
  for (int i = threadIdx.x; i < W_BYTES / 16; i += WG) reinterpret_cast<u32x4*>(lds)[i] = rnd(977u * stage + i);
  __syncthreads();
  const int links = a.stages - 1;
  unsigned char* ring_in = stage > 0 ? a.ring + ((size_t)(chain * links + stage - 1) * a.R) * TILE_BYTES : nullptr;
  unsigned char* ring_out = stage < links ? a.ring + ((size_t)(chain * links + stage) * a.R) * TILE_BYTES : nullptr;
  unsigned* ready_in = stage > 0 ? a.ready + (size_t)(chain * links + stage - 1) * 16 : nullptr;
  unsigned* acked_in = stage > 0 ? a.acked + (size_t)(chain * links + stage - 1) * 16 : nullptr;
  unsigned* ready_out = stage < links ? a.ready + (size_t)(chain * links + stage) * 16 : nullptr;
  unsigned* acked_out = stage < links ? a.acked + (size_t)(chain * links + stage) * 16 : nullptr;
  const bool handoff = a.mode != 0;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(ring_in ? ring_in : a.ring, 0, a.R * TILE_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(ring_out ? ring_out : a.ring, 0, a.R * TILE_BYTES, 0x00020000);

  u32x4 X[16];
  for (int ks = 0; ks < 16; ++ks) X[ks] = rnd(131u * b + 16 * threadIdx.x + ks);
  float keep = 0.f;
  bool ok = true;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < a.tiles && ok; ++t) {
    const int slot = t % a.R;
    // ---- input tile
    if (handoff && stage > 0) {
      if (threadIdx.x == 0) {
        ok = spin_until_ge(ready_in, (unsigned)t + 1, a.timeout);
        if (a.mode == 2) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      }
      __syncthreads();
      const int off = slot * TILE_BYTES + wv * 16 * PIECE + lane * 16;
      const unsigned char* src = ring_in + off;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (a.mode == 1)
          X[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, ks * PIECE, 16);  // aux 16 = sc1: L2 / memory-side, never this CU's L1
        else
          X[ks] = *reinterpret_cast<const u32x4*>(src + ks * PIECE);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave has its pieces: the slot may be overwritten
      if (threadIdx.x == 0) __hip_atomic_store(acked_in, (unsigned)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the layer: 8 output tiles x 16 k-steps, accumulator -> packed operand of the next layer
    u32x4 Y[16];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const u32x4 A = *reinterpret_cast<const u32x4*>(lds + (size_t)(f * 16 + ks) * PIECE + lane * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, X[ks]), acc, 0, 0, 0);
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int q = 0; q < 4; ++q) Y[2 * f + mh][q] = pack2_relu(acc[8 * mh + 2 * q], acc[8 * mh + 2 * q + 1]);
    }
    // ---- output tile
    if (handoff && stage < links) {
      if (t >= a.R) {  // back-pressure: the consumer must have taken tile t - R out of this slot
        if (threadIdx.x == 0) ok = spin_until_ge(acked_out, (unsigned)(t - a.R) + 1, a.timeout);
        __syncthreads();
      }
      const int off = slot * TILE_BYTES + wv * 16 * PIECE + lane * 16;
      unsigned char* dst = ring_out + off;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (a.mode == 1)
          __builtin_amdgcn_raw_buffer_store_b128(Y[ks], rs_out, off, ks * PIECE, 16);  // sc1: write-through
        else
          *reinterpret_cast<u32x4*>(dst + ks * PIECE) = Y[ks];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
      __syncthreads();
      if (threadIdx.x == 0) {
        if (a.mode == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_store(ready_out, (unsigned)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    keep += __uint_as_float(Y[3][1] << 16) + __uint_as_float(Y[12][2] << 16);
    if (!handoff || stage == 0) {  // made-up inputs: a new tile per step
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) X[ks][0] ^= Y[ks][1] & 0x00010001u;
    }
    ok = __shfl(ok ? 1 : 0, 0) != 0;  // (lane 0 of each wave saw the barrier-shared state; keep the loop condition wave-uniform)
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    ok = s_ok != 0;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) a.cycles[b] = t1 - t0;
  a.sink[(size_t)b * WG + threadIdx.x] = keep;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 2000;
  const int R = 4;
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  const int lds_bytes = W_BYTES + 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const int blocks = 256;
  float* sink;
  unsigned long long* cycles;
  unsigned *ready, *acked, *timeout;
  unsigned char* ring;
  const int max_links = 32 * 8;  // generous
  CHECK(hipMalloc(&sink, (size_t)blocks * WG * 4));
  CHECK(hipMalloc(&cycles, blocks * 8));
  CHECK(hipMalloc(&ready, max_links * 64));
  CHECK(hipMalloc(&acked, max_links * 64));
  CHECK(hipMalloc(&timeout, 4));
  CHECK(hipMalloc(&ring, (size_t)max_links * R * TILE_BYTES));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double base_ns[9] = {};
  for (int only_xcd : {0, -1}) {
    for (int stages : {2, 4, 8}) {
      for (int mode : {0, 1, 2}) {
        Args a;
        a.ring = ring; a.ready = ready; a.acked = acked; a.timeout = timeout; a.sink = sink; a.cycles = cycles;
        a.stages = stages; a.R = R; a.tiles = tiles; a.mode = mode; a.only_xcd = only_xcd; a.chains_per_xcd = 32 / stages;
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {  // the first run warms up (LDS opt-in, clocks)
          CHECK(hipMemset(ready, 0, max_links * 64));
          CHECK(hipMemset(acked, 0, max_links * 64));
          CHECK(hipMemset(timeout, 0, 4));
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(WG), lds_bytes, 0, a);
          CHECK(hipGetLastError());
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned to = 0;
        CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        const int xcds = only_xcd >= 0 ? 1 : 8;
        const int chains = xcds * a.chains_per_xcd, links = stages - 1;
        const double ns_tile = ms * 1e6 / tiles;
        const double gbs_chain_link = TILE_BYTES / ns_tile;  // GB/s through ONE link (written once, read once)
        if (mode == 0) base_ns[stages] = ns_tile;
        printf("xcds=%d stages=%d mode=%d%s: %.1f ns/tile  hand-off %.1f GB/s per link, %.1f GB/s per XCD (%d links), %.2f TB/s chip-wide"
               "  MFMA-busy vs mode 0: %.2f  (pure MFMA issue at 2.1 GHz: %.0f ns/tile)\n",
               xcds, stages, mode, to ? " TIMEOUT(invalid)" : "", ns_tile, mode ? gbs_chain_link : 0.0,
               mode ? gbs_chain_link * a.chains_per_xcd * links : 0.0, a.chains_per_xcd * links,
               mode ? gbs_chain_link * chains * links / 1e3 : 0.0, base_ns[stages] / ns_tile, 2 * 128 * 32 / 2.1);
      }
    }
  }
  return 0;
}
