// bf16_common.h -- layout of the bf16 weight STREAM used by the bf16-MLP variant of the field query (cfg3 of
// BASELINE.json: "bf16 MLP / fp32 composite").  MI355X / gfx950 only.
//
// Arithmetic of the variant (restated by tests/test_gpu_bf16.py's torch emulation): every linear layer of the field
// MLP (nerf.py:78-99) multiplies bf16-rounded weights with bf16-rounded inputs, accumulates in fp32 starting from the
// fp32 bias, and hands its (ReLU'd) output to the next layer rounded to bf16 (round-to-nearest-even).  The positional
// encodings are computed in fp32 exactly as in the fp32 path and rounded to bf16 once.  sigma = |.| and the colour sigmoid are
// fp32 on the fp32 accumulators; everything outside the MLP (rays, depths, compositing, resampling, sort) stays fp32.
//
// The image holds one v_mfma_f32_32x32x16_bf16 A fragment (32 output features x 16 inputs, 8 bf16 = 16 bytes per lane,
// 1 KiB per wave) per MFMA, in the exact order a wave consumes them: layer by layer, output tile by output tile,
// k-step by k-step.  Lane (i, h) of fragment (f, ks) holds W[32f + i][k] for the 8 inputs
//     k = 16ks + 4h + {0,1,2,3} and 16ks + 8 + 4h + {0,1,2,3}
// which is where the 32x32 fp32 accumulator layout leaves them: lane (sample j, half h) of an accumulator tile holds
// features 8g + 4h + r of sample j (register 4g + r), so registers 0..7 / 8..15 of tile t, converted pairwise to bf16,
// ARE the B operands of k-steps 2t / 2t + 1 of the next layer -- no shuffle, no LDS round trip for activations.
// In front of the fragments sits one 16-KiB block with the fp32 biases per output tile.
#pragma once
#include "common.h"

namespace nerf {

constexpr int BF_FRAG_BYTES = 1024;
constexpr int BF_CHUNK = 16;              // fragments per LDS ring slot (16 KiB)
constexpr int BF_BIAS_BYTES = 16384;      // bias block (78 tiles x 32 floats used)

// stream segments: first fragment, output tiles, k-steps.  point_info (no activation, nerf.py:117) is folded into dir_info's feature
// columns (common.h SEG_FOLD): W_fold = W_dir[:, 24:] W_pi is formed in fp32 (k_fold_weights) and rounded to bf16 ONCE; the sigma
// head is a tile of its own on h7.
constexpr int BFS_L0 = 0;      // 8 tiles x 4   gamma_p (60 -> 64)
constexpr int BFS_L1 = 32;     // 8 x 16, likewise L2, L3
constexpr int BFS_L4 = 416;    // 8 x (16 hidden + 4 gamma_p)   (nerf.py:109: hidden first)
constexpr int BFS_L5 = 576;    // 8 x 16, likewise L6, L7
constexpr int BFS_SIG = 960;   // 1 x 16: row 0 = sigma_layer (on h7)
constexpr int BFS_DIR = 976;   // 4 x (2 gamma_d (24 -> 32) + 16 h7 through W_fold)   (nerf.py:117: direction first)
constexpr int BFS_COL = 1048;  // 1 x 8: rows 0..2 = color_layer
constexpr int BF_NFRAG = 1056;
constexpr int BF_NCHUNK = BF_NFRAG / BF_CHUNK;  // 66
static_assert(BF_NCHUNK * BF_CHUNK == BF_NFRAG, "stream must be whole chunks");
constexpr size_t BF_IMAGE_BYTES = (size_t)BF_BIAS_BYTES + (size_t)BF_NFRAG * BF_FRAG_BYTES;

// ---- backward (dX chain) stream: transposed weights, consumed from the colour head back to layer 0 --------------------
//  COLT  4 tiles x 4 k-steps (only k-step 0, inputs 0..2 = dz, are non-zero)       d c    = W_color^T dz
//  FOLDT 8 x 9: 8 k-steps W_fold^T dpre_dir + 1 k-step w_sigma (input slot 3 = dspre)   d h7
//  L7T, L6T, L5T, L4T (hidden columns), L3T, L2T, L1T: 8 x 16 each
//  fine pass only, d gamma_p:  2 tiles x (16 k-steps of W_0^T (input dpre0) + 16 k-steps of W_4[:, 256:]^T (input dpre4, re-read))
// One image serves both passes: the coarse pass stops after L1T.
constexpr int BBS_COLT = 0, BBS_FOLDT = 16, BBS_L7T = 88, BBS_L4T = 472, BBS_L3T = 600;
constexpr int BBS_G0T = 984;
constexpr int BBC_NFRAG = 984, BBF_NFRAG = 1048;
constexpr int BBC_NCHUNK = (BBC_NFRAG + BF_CHUNK - 1) / BF_CHUNK, BBF_NCHUNK = (BBF_NFRAG + BF_CHUNK - 1) / BF_CHUNK;
constexpr size_t BB_IMAGE_BYTES = (size_t)BF_BIAS_BYTES + (size_t)BBF_NCHUNK * BF_CHUNK * BF_FRAG_BYTES;

// ---- training buffers in FRAGMENT layout: per wave block (32 consecutive samples of a pass) and tensor, ks pieces of
// 1 KiB = what the wave's 64 lanes hold as one packed B operand (lane (j, h): sample j, features 16ks + 4h + {0..3, 8..11}).
// Tensor-major: tensor t starts at wb_tot * 1024 * cum_ks(t); piece (wb, ks) of it at ((wb * ks_t) + ks) * 1024.
// saved by the forward (point_info's output is not: folded, no weight gradient needs it):
constexpr int BS_GP = 0, BS_H0 = 1, BS_C = 9, BS_GD = 10, NBS = 11;
__host__ __device__ constexpr int bs_ks(int t) { return t == BS_GP ? 4 : t < BS_C ? 16 : t == BS_C ? 8 : 2; }
__host__ __device__ constexpr int bs_cum(int t) { int o = 0; for (int i = 0; i < t; ++i) o += bs_ks(i); return o; }
constexpr int BS_TOTAL_KS = bs_cum(NBS);  // 142 KiB per wave block
// written by the backward chain (pre-activation gradients = A operands of the dW GEMMs):
constexpr int BG_L0 = 0, BG_D = 8, BG_Z = 9, NBG = 10;  // BG_Z: features 0..2 = dz (colour), 3 = dspre (sigma)
__host__ __device__ constexpr int bg_ks(int t) { return t < BG_D ? 16 : t == BG_D ? 8 : 2; }  // BG_Z: 2nd piece = zeros
__host__ __device__ constexpr int bg_cum(int t) { int o = 0; for (int i = 0; i < t; ++i) o += bg_ks(i); return o; }
constexpr int BG_TOTAL_KS = bg_cum(NBG);  // 138
// ReLU "alive" masks: u16 [9 layers: h0..h7, c][wb_tot][64 lanes][8 tiles], bit 15-r = accumulator register r >= +0
constexpr int BM_LAYERS = 9;

// bias tiles (32 floats each) in the bias block
constexpr int BFB_L0 = 0, BFB_SIGMA = 64, BFB_DIR = 65, BFB_COL = 69, BF_NBIAS_TILES = 70;  // BFB_DIR: b_dir + W_dir[:, 24:] b_pi


}  // namespace nerf
