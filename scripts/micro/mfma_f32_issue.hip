// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate with 1 vs 2 waves per SIMD (registers only, no memory).
// hipcc --offload-arch=gfx950 -O3 mfma_f32_issue.hip -o mfma_f32_issue && ./mfma_f32_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / NACC * 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, const char* label) {
  float* out; hipMalloc(&out, 4 * 256 * 256 * 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 100, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma = (double)256 * wgs_per_cu * 4 * iters * (32 / NACC * 4) * NACC;
  double flops = mfma * 32 * 32 * 2 * 2;
  printf("%-34s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", label, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mfma / 1024));
  hipFree(out);
}
int main() {
  run<8>(1, "1 wave/SIMD, 8 accumulators, 128 MFMA/iter");
  run<8>(2, "2 waves/SIMD, 8 accumulators, 128 MFMA/iter");
  run<4>(1, "1 wave/SIMD, 4 accumulators");
  run<4>(2, "2 waves/SIMD, 4 accumulators");
  run<4>(4, "4 waves/SIMD, 4 accumulators");
  run<1>(1, "1 wave/SIMD, 1 accumulator");
  run<1>(2, "2 waves/SIMD, 1 accumulator");
  run<2>(2, "2 waves/SIMD, 2 accumulators");
  return 0;
}
