// bf16_weights.h -- element (fragment, row, input) of the bf16 forward weight stream of bf16_common.h and of its bias block, shared by
// the packers of the 32x32x16 bf16 image (field_fwd_bf16.hip) and of the split-fp32 image (field_fwd_split.hip).
#pragma once
#include "bf16_common.h"

namespace nerf {

__device__ __forceinline__ float bf_weight(const Weights24& w, const float* __restrict__ fold, int frag, int i, int kk /* 0..15 inside the k-step */, int h) {
  (void)h;
  if (frag < BFS_L1) {  // L0
    const int f = frag / 4, ks = frag % 4, k = 16 * ks + kk;
    return k < POINT_DIM ? w.p[0][(size_t)(32 * f + i) * POINT_DIM + k] : 0.f;
  }
  if (frag < BFS_L4) {  // L1..L3
    const int r = frag - BFS_L1, l = 1 + r / 128, q = r % 128, f = q / 16, ks = q % 16;
    return w.p[2 * l][(size_t)(32 * f + i) * WIDTH + 16 * ks + kk];
  }
  if (frag < BFS_L5) {  // L4: [256][316] = cat(hidden, gamma_p)
    const int q = frag - BFS_L4, f = q / 20, ks = q % 20, k = 16 * ks + kk;
    return (k < WIDTH + POINT_DIM) ? w.p[8][(size_t)(32 * f + i) * (WIDTH + POINT_DIM) + k] : 0.f;
  }
  if (frag < BFS_SIG) {  // L5..L7
    const int r = frag - BFS_L5, l = 5 + r / 128, q = r % 128, f = q / 16, ks = q % 16;
    return w.p[2 * l][(size_t)(32 * f + i) * WIDTH + 16 * ks + kk];
  }
  if (frag < BFS_DIR) {  // sigma row on h7
    const int ks = frag - BFS_SIG, k = 16 * ks + kk;
    return i == 0 ? w.p[W_SIGMA][k] : 0.f;
  }
  if (frag < BFS_COL) {  // dir_info: gamma_d columns of W_dir (24 -> 32), then W_fold = W_dir[:, 24:] W_pi on h7
    const int q = frag - BFS_DIR, f = q / 18, ks = q % 18, k = 16 * ks + kk;
    if (ks < 2) return k < DIR_DIM ? w.p[W_DIR][(size_t)(32 * f + i) * (WIDTH + DIR_DIM) + k] : 0.f;
    return fold[HALF + (size_t)(32 * f + i) * WIDTH + (k - 32)];
  }
  const int ks = frag - BFS_COL, k = 16 * ks + kk;  // colour head
  return i < 3 ? w.p[W_COLOR][(size_t)i * HALF + k] : 0.f;
}

__device__ __forceinline__ float bf_bias(const Weights24& w, const float* __restrict__ fold, int tile, int i) {
  if (tile < BFB_SIGMA) return w.p[2 * (tile / 8) + 1][32 * (tile % 8) + i];
  if (tile == BFB_SIGMA) return i == 0 ? w.p[B_SIGMA][0] : 0.f;
  if (tile < BFB_COL) return w.p[B_DIR][32 * (tile - BFB_DIR) + i] + fold[32 * (tile - BFB_DIR) + i];  // + W_dir[:, 24:] b_pi
  return i < 3 ? w.p[B_COLOR][i] : 0.f;
}


}  // namespace nerf
