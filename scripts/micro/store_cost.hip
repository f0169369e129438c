// Microbenchmark: what do the activation stores of the training kernels cost inside a v_mfma_f32_32x32x2_f32 stream?
// One wave per SIMD on every CU (as k_field_fwd_reg<SAVE>), blocks of 32 MFMAs, 4 x 16-byte stores (1 KiB per instruction) or
// 16 x 4-byte stores behind each block, streaming to fresh memory.  Patterns:
//   rows   lane (j, h) -> row j, 16 bytes at column 8g + 4h: the C layout of the MFMA written as it is (32 rows x 32 B per instruction)
//   flat   lane l -> 16 bytes at l * 16: one contiguous KiB per instruction (a tile-blocked layout)
//   8rows  lane l -> row l / 8, 16 bytes at (l % 8) * 16: eight full 128-byte lines per instruction (rows transposed through LDS first)
//   dword  lane (j, h) -> 4 bytes, row r + h, column j: a transposed product's layout (two full lines per instruction, 16 instructions)
// hipcc --offload-arch=gfx950 -O3 store_cost.hip -o store_cost && ./store_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc, a, b) "v_mfma_f32_32x32x2_f32 a[" acc "], " a ", " b ", a[" acc "]\n\t"
#define BLOCK_ROT(b) MF("0:15", "v4", b) MF("16:31", "v5", b) MF("32:47", "v6", b) MF("48:63", "v7", b) MF("64:79", "v8", b) MF("80:95", "v9", b) MF("96:111", "v10", b) MF("112:127", "v11", b)
#define MFMA32 BLOCK_ROT("v12") BLOCK_ROT("v12") BLOCK_ROT("v12") BLOCK_ROT("v12")
#define ST4(p, off) "global_store_dwordx4 " p ", v[24:27], off offset:" off "\n\t"
#define ST1(p, off) "global_store_dword " p ", v24, off offset:" off "\n\t"
#define DW(off) "ds_write_b128 %4, v[24:27] offset:" off "\n\t"
#define DR(off) "ds_read_b128 v[28:31], %4 offset:" off "\n\t"

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, char* buf, int iters) {
  __shared__ float sh[2048];
  sh[threadIdx.x] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  asm volatile(
      "v_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\tv_mov_b32 v6, 1.0\n\tv_mov_b32 v7, 1.0\n\t"
      "v_mov_b32 v8, 1.0\n\tv_mov_b32 v9, 1.0\n\tv_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\tv_mov_b32 v12, 0.5\n\t"
      "v_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 1.0\n\t" ::
          : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v24", "v25", "v26", "v27");
  // 32 rows x 1 KiB per 8 iterations (one 128-byte tile column per iteration), like the saves of one layer
  char* wave_base = buf + (size_t)blockIdx.x * (size_t)((iters + 7) / 8) * 32768;
  const unsigned lds_addr = (unsigned)(lane * 16);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    char* blk = wave_base + (size_t)(it >> 3) * 32768 + (it & 7) * 128;
    char* p0 = blk, *p1 = blk, *p2 = blk, *p3 = blk;
    if (MODE == 1 || MODE == 0) { p0 = blk + j * 1024 + h * 16; p1 = p0; p2 = p0; p3 = p0; }
    if (MODE == 2) { p0 = wave_base + (size_t)it * 4096 + lane * 16; }
    if (MODE == 3 || MODE == 5) { p0 = blk + (lane >> 3) * 1024 + (lane & 7) * 16; p1 = p0 + 8192; p2 = p0 + 16384; p3 = p0 + 24576; }
    if (MODE == 4) { p0 = blk + h * 1024 + j * 4 + 4096; p1 = p0 + 8192; p2 = p0 + 16384; p3 = p0 + 24576; }
    if (MODE == 0) asm volatile(MFMA32 :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory");
    if (MODE == 1) asm volatile(MFMA32 ST4("%0", "0") ST4("%0", "32") ST4("%0", "64") ST4("%0", "96") :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory");
    if (MODE == 2) asm volatile(MFMA32 ST4("%0", "0") ST4("%0", "1024") ST4("%0", "2048") ST4("%0", "3072") :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory");
    if (MODE == 3) asm volatile(MFMA32 ST4("%0", "0") ST4("%1", "0") ST4("%2", "0") ST4("%3", "0") :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory");
    if (MODE == 4)
      asm volatile(MFMA32 ST1("%0", "-4096") ST1("%0", "-2048") ST1("%0", "0") ST1("%0", "2048") ST1("%1", "-4096") ST1("%1", "-2048") ST1("%1", "0") ST1("%1", "2048")
                       ST1("%2", "-4096") ST1("%2", "-2048") ST1("%2", "0") ST1("%2", "2048") ST1("%3", "-4096") ST1("%3", "-2048") ST1("%3", "0") ST1("%3", "2048")
                   :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory");
    // the LDS round trip of a transposing store: 4 ds_write_b128 + 4 ds_read_b128 (the reads' data is not what is stored here;
    // only the instruction cost matters) + the 8-full-lines stores
    if (MODE == 5)
      asm volatile(MFMA32 DW("0") DW("1024") DW("2048") DW("3072") DR("0") DR("1024") DR("2048") DR("3072")
                       ST4("%0", "0") ST4("%1", "0") ST4("%2", "0") ST4("%3", "0") "s_waitcnt lgkmcnt(0)\n\t"
                   :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(lds_addr) : "memory", "v28", "v29", "v30", "v31");
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* label, char* buf, double base = 0) {
  unsigned long long* out;
  hipMalloc(&out, 8 * 1024);
  const int iters = 1024;
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, buf, 16);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, buf, iters);
  hipDeviceSynchronize();
  unsigned long long h[1024];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%-64s %8.2f s_memtime ticks per block of 32 MFMAs\n", label, s / 1024 / iters);
  hipFree(out);
}

int main() {
  char* buf;
  const size_t bytes = (size_t)1024 * 128 * 32768 + (1 << 20);  // 4 GiB
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  run<0>("32 MFMAs, no stores", buf);
  run<1>("+ 4 dwordx4 stores, rows (32 rows x 32 B per instruction)", buf);
  run<2>("+ 4 dwordx4 stores, flat (1 contiguous KiB per instruction)", buf);
  run<3>("+ 4 dwordx4 stores, 8rows (8 full lines per instruction)", buf);
  run<4>("+ 16 dword stores, 2 full lines per instruction", buf);
  run<5>("+ 4 ds_write_b128 + 4 ds_read_b128 + 4 dwordx4 stores 8rows", buf);
  run<0>("32 MFMAs, no stores (again)", buf);
  hipFree(buf);
  return 0;
}
