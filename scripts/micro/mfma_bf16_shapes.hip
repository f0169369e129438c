// Microbenchmark: sustained bf16 MFMA rate of the 32x32x16 and the 16x16x32 form on RANDOM operands (the clock the chip
// holds under an MFMA load depends on the data), 8 waves per CU = 2 per SIMD, operands in registers, dependent chains of 16
// as in field_fwd_bf16.hip.  hipcc --offload-arch=gfx950 -O3 mfma_bf16_shapes.hip -o x && ./x
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ u32x4 rnd(unsigned seed) {  // 8 bf16 values in [-1, 1)
  u32x4 r;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = hash(seed * 4 + i);
    const unsigned lo = 0x3f80u | (h & 0x7fu) | ((h >> 7) & 1u) << 15, hi = 0x3f80u | ((h >> 8) & 0x7fu) | ((h >> 15) & 1u) << 15;
    r[i] = lo | hi << 16;
  }
  return r;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd(threadIdx.x * 16 + i); b[i] = rnd(threadIdx.x * 16 + 8 + i); }
  f32x16 acc = {0};
  f32x4 c0 = {0}, c1 = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const bf16x8 A = __builtin_bit_cast(bf16x8, a[u & 7]), B = __builtin_bit_cast(bf16x8, b[(u * 3) & 7]);
      if (MODE == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
      } else {
        const bf16x8 B2 = __builtin_bit_cast(bf16x8, b[(u * 3 + 1) & 7]);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, c1, 0, 0, 0);
      }
    }
    if (MODE == 0) { for (int r = 0; r < 16; ++r) acc[r] *= 1e-3f; } else { for (int r = 0; r < 4; ++r) { c0[r] *= 1e-3f; c1[r] *= 1e-3f; } }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int r = 0; r < 4; ++r) s += c0[r] + c1[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* label) {
  float* out;
  hipMalloc(&out, 4 * 512 * 1024);
  const int iters = 20000, wgs = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, out, 200);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 8 * iters * 16 * (MODE == 0 ? 32768.0 : 2 * 16384.0);
  printf("%-28s %.1f TFLOP/s  (%.2f ms)\n", label, flop / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
}

int main() {
  run<0>("v_mfma_f32_32x32x16_bf16");
  run<1>("v_mfma_f32_16x16x32_bf16 x2");
  run<0>("v_mfma_f32_32x32x16_bf16");
  run<1>("v_mfma_f32_16x16x32_bf16 x2");
  return 0;
}
