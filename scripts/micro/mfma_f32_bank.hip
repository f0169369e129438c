// Microbenchmark: does the VGPR bank of the A / B operands change the issue rate of v_mfma_f32_32x32x2_f32?
// 8 independent accumulators (a[0:127]) as in k_field_fwd_reg; A and B registers chosen by hand.
// hipcc --offload-arch=gfx950 -O3 mfma_f32_bank.hip -o mfma_f32_bank && ./mfma_f32_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc, a, b) "v_mfma_f32_32x32x2_f32 a[" acc "], " a ", " b ", a[" acc "]\n\t"
#define EIGHT(a, b) MF("0:15", a, b) MF("16:31", a, b) MF("32:47", a, b) MF("48:63", a, b) MF("64:79", a, b) MF("80:95", a, b) MF("96:111", a, b) MF("112:127", a, b)
// four A registers per B as in the kernel (float4 fragment: consecutive registers = all four banks)
#define BLOCK_ROT(b) MF("0:15", "v4", b) MF("16:31", "v5", b) MF("32:47", "v6", b) MF("48:63", "v7", b) MF("64:79", "v8", b) MF("80:95", "v9", b) MF("96:111", "v10", b) MF("112:127", "v11", b)

// one MFMA followed by N independent VALU instructions
#define V1 "v_max_f32 v20, v21, v22\n\t"
#define MFV(acc, a, b, fill) MF(acc, a, b) fill
#define ROT_FILL(b, fill) MFV("0:15", "v4", b, fill) MFV("16:31", "v5", b, fill) MFV("32:47", "v6", b, fill) MFV("48:63", "v7", b, fill) MFV("64:79", "v8", b, fill) MFV("80:95", "v9", b, fill) MFV("96:111", "v10", b, fill) MFV("112:127", "v11", b, fill)
#define RD "ds_read_b128 v[24:27], v23\n\t"
#define ACR "v_accvgpr_read_b32 v20, a200\n\t"

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters) {
  __shared__ float sh[1024];
  sh[threadIdx.x] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  asm volatile(
      "v_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\tv_mov_b32 v6, 1.0\n\tv_mov_b32 v7, 1.0\n\t"
      "v_mov_b32 v8, 1.0\n\tv_mov_b32 v9, 1.0\n\tv_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\tv_mov_b32 v12, 0.5\n\tv_mov_b32 v13, 0.5\n\t" ::
          : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
  for (int i = 0; i < 128; ++i) asm volatile("" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile(EIGHT("v4", "v8") EIGHT("v4", "v8") EIGHT("v4", "v8") EIGHT("v4", "v8")::: "memory");      // A, B same bank (4, 8)
    if (MODE == 1) asm volatile(EIGHT("v4", "v9") EIGHT("v4", "v9") EIGHT("v4", "v9") EIGHT("v4", "v9")::: "memory");      // different banks
    if (MODE == 2) asm volatile(BLOCK_ROT("v12") BLOCK_ROT("v12") BLOCK_ROT("v12") BLOCK_ROT("v12")::: "memory");        // kernel-like: A rotates over banks
    if (MODE == 3) asm volatile(EIGHT("v4", "v4") EIGHT("v4", "v4") EIGHT("v4", "v4") EIGHT("v4", "v4")::: "memory");      // same register
    if (MODE == 4) asm volatile(ROT_FILL("v12", V1 V1) ROT_FILL("v12", V1 V1) ROT_FILL("v12", V1 V1) ROT_FILL("v12", V1 V1)::: "memory", "v20");
    if (MODE == 5) asm volatile(ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1)::: "memory", "v20");
    if (MODE == 6) asm volatile(ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1) ROT_FILL("v12", V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1 V1)::: "memory", "v20");
    if (MODE == 7) asm volatile("v_mov_b32 v23, 0\n\t" ROT_FILL("v12", RD) ROT_FILL("v12", RD) ROT_FILL("v12", RD) ROT_FILL("v12", RD) "s_waitcnt lgkmcnt(0)\n\t" ::: "memory", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 8) asm volatile(ROT_FILL("v12", ACR ACR) ROT_FILL("v12", ACR ACR) ROT_FILL("v12", ACR ACR) ROT_FILL("v12", ACR ACR)::: "memory", "v20");
    if (MODE == 9) asm volatile(ROT_FILL("v12", "s_waitcnt lgkmcnt(7)\n\t") ROT_FILL("v12", "s_waitcnt lgkmcnt(7)\n\t") ROT_FILL("v12", "s_waitcnt lgkmcnt(7)\n\t") ROT_FILL("v12", "s_waitcnt lgkmcnt(7)\n\t")::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* label) {
  unsigned long long* out;
  hipMalloc(&out, 8 * 1024);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, 10);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, iters);
  hipDeviceSynchronize();
  unsigned long long h[1024];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%-42s %.2f ticks of s_memtime per MFMA (1 wave per SIMD, 1024 waves)\n", label, s / 1024 / iters / 32);
  hipFree(out);
}

int main() {
  run<0>("A, B in the same VGPR bank");
  run<1>("A, B in different banks");
  run<2>("A rotating over 8 registers, one B");
  run<3>("A = B (same register)");
  run<4>("+ 2 VALU (v_max_f32) per MFMA");
  run<5>("+ 8 VALU per MFMA");
  run<6>("+ 16 VALU per MFMA");
  run<7>("+ 1 ds_read_b128 per MFMA");
  run<8>("+ 2 v_accvgpr_read per MFMA");
  run<9>("+ 1 s_waitcnt lgkmcnt(7) per MFMA");
  return 0;
}
