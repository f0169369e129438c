// bf16_weights.h -- element (fragment, row, input) of the bf16 weight streams of bf16_common.h (32x32x16 forward image and its bias block,
// 16x16x32 inference image, transposed backward image), shared by their packers (field_fwd_bf16.hip, field_fwd_bf16x.hip, field_bwd_bf16.hip,
// field_fwd_split.hip) and by the one-launch preparation kernel (prep_bf16.hip).
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


// ---- the 16x16x32 inference image (field_fwd_bf16x.hip) ----
// stream segments (16-row tiles x 32-wide k-steps), first fragment of each
constexpr int BXS_L0 = 0;      // 16 tiles x 2
constexpr int BXS_L1 = 32;     // 16 x 8, likewise L2, L3
constexpr int BXS_L4 = 416;    // 16 x (8 hidden + 2 gamma_p)
constexpr int BXS_L5 = 576;    // 16 x 8, likewise L6, L7
constexpr int BXS_SIG = 960;   // 1 x 8: row 0 = sigma_layer (on h7)
constexpr int BXS_DIR = 968;   // 8 x (1 gamma_d + 8 h7 through W_fold: point_info folded into dir_info, bf16_common.h)
constexpr int BXS_COL = 1040;  // 1 x 4
constexpr int BX_NFRAG = 1044;
constexpr int BX_NCHUNK = (BX_NFRAG + BF_CHUNK - 1) / BF_CHUNK;  // 66 (the last chunk is padded)
static_assert((size_t)BF_BIAS_BYTES + (size_t)BX_NCHUNK * BF_CHUNK * BF_FRAG_BYTES <= BF_IMAGE_BYTES, "shares the workspace region of the 32x32x16 image");
// fragment (tile T, k-step s), lane (i, q), slot j = W[16T + i][32s + 16 (j >> 2) + 4q + (j & 3)]
__device__ __forceinline__ float bx_weight(const Weights24& w, const float* __restrict__ fold, int frag, int i, int kk /* input feature inside the k-step */) {
  if (frag < BXS_L1) {  // L0
    const int T = frag / 2, s = frag % 2, k = 32 * s + kk;
    return k < POINT_DIM ? w.p[0][(size_t)(16 * T + i) * POINT_DIM + k] : 0.f;
  }
  if (frag < BXS_L4) {  // L1..L3
    const int r = frag - BXS_L1, l = 1 + r / 128, x = r % 128, T = x / 8, s = x % 8;
    return w.p[2 * l][(size_t)(16 * T + i) * WIDTH + 32 * s + kk];
  }
  if (frag < BXS_L5) {  // L4: [256][316] = cat(hidden, gamma_p)
    const int x = frag - BXS_L4, T = x / 10, s = x % 10, k = 32 * s + kk;
    return (k < WIDTH + POINT_DIM) ? w.p[8][(size_t)(16 * T + i) * (WIDTH + POINT_DIM) + k] : 0.f;
  }
  if (frag < BXS_SIG) {  // L5..L7
    const int r = frag - BXS_L5, l = 5 + r / 128, x = r % 128, T = x / 8, s = x % 8;
    return w.p[2 * l][(size_t)(16 * T + i) * WIDTH + 32 * s + kk];
  }
  if (frag < BXS_DIR) {  // the sigma tile on h7
    const int s = frag - BXS_SIG, k = 32 * s + kk;
    return i == 0 ? w.p[W_SIGMA][k] : 0.f;
  }
  if (frag < BXS_COL) {  // dir_info: gamma_d columns of W_dir (24 -> 32), then W_fold = W_dir[:, 24:] W_pi on h7
    const int x = frag - BXS_DIR, T = x / 9, s = x % 9;
    if (s == 0) return kk < DIR_DIM ? w.p[W_DIR][(size_t)(16 * T + i) * (WIDTH + DIR_DIM) + kk] : 0.f;
    return fold[HALF + (size_t)(16 * T + i) * WIDTH + 32 * (s - 1) + kk];
  }
  if (frag < BX_NFRAG) {  // colour head
    const int s = frag - BXS_COL, k = 32 * s + kk;
    return i < 3 ? w.p[W_COLOR][(size_t)i * HALF + k] : 0.f;
  }
  return 0.f;  // padding of the last chunk
}

// ---- the transposed image of the backward chain (field_bwd_bf16.hip; segments: bf16_common.h BBS_*) ----
// fragment (f, ks), lane (i, h), slot s  =  W[out = 16ks + 4h + (s&3) + 8(s>>2)][in = 32f + i]
__device__ __forceinline__ float bb_weight(const Weights24& w, const float* __restrict__ fold, int frag, int i, int kk) {
  if (frag < BBS_FOLDT) {  // COLT: d c = W_color^T dz
    const int f = frag / 4, ks = frag % 4, k = 16 * ks + kk;
    return k < 3 ? w.p[W_COLOR][(size_t)k * HALF + 32 * f + i] : 0.f;
  }
  if (frag < BBS_L7T) {  // FOLDT: 8 k-steps of W_fold^T (W_fold = W_dir[:, 24:] W_pi, [128][256]), then the sigma step (input slot 3)
    const int q = frag - BBS_FOLDT, f = q / 9, ks = q % 9, k = 16 * ks + kk;
    if (ks < 8) return fold[HALF + (size_t)k * WIDTH + 32 * f + i];
    return kk == 3 ? w.p[W_SIGMA][32 * f + i] : 0.f;
  }
  if (frag < BBS_G0T) {  // L7T .. L1T (layer 4: hidden columns of the [256][316] matrix)
    const int r = frag - BBS_L7T, l = 7 - r / 128, q = r % 128, f = q / 16, ks = q % 16;
    const int ld = (l == 4) ? WIDTH + POINT_DIM : WIDTH;
    return w.p[2 * l][(size_t)(16 * ks + kk) * ld + 32 * f + i];
  }
  if (frag < BBF_NFRAG) {  // d gamma_p, fine pass: tile f = 16 k-steps of W_0^T (input dpre0), then 16 k-steps of W_4[:, 256:]^T (input dpre4)
    const int q = frag - BBS_G0T, f = q / 32, ks = q % 32, col = 32 * f + i;
    if (col >= POINT_DIM) return 0.f;
    if (ks < 16) return w.p[0][(size_t)(16 * ks + kk) * POINT_DIM + col];
    return w.p[8][(size_t)(16 * (ks - 16) + kk) * (WIDTH + POINT_DIM) + WIDTH + col];
  }
  return 0.f;  // padding up to whole chunks
}


}  // namespace nerf
