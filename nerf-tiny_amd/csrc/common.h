// common.h -- shared constants, workspace layout and kernel argument blocks of libnerf_hip.so.
// MI355X / gfx950 only.  Reference citations are to /root/reference/nerf.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nerf {

constexpr int WIDTH = 256;      // nerf.py:76
constexpr int POINT_DIM = 60;   // 3 * 2 * L_point (nerf.py:127, :103)
constexpr int DIR_DIM = 24;     // 3 * 2 * L_dir
constexpr int L_POINT = 10;
constexpr int L_DIR = 4;
constexpr int HALF = 128;       // width of the colour branch (nerf.py:98)
constexpr int TM = 64;          // samples per workgroup tile of the field kernels
// Status region = 64 u32 words.  Word 0: the flags of the last forward in the LEGACY scheme (zeroed by the call's first kernel, OR-ed by later
// ones).  A forward whose first kernel is also the one that sets flags (the ray-pair kernel of small bf16 inference batches: one launch per
// call) cannot zero anything first; it uses the STAMPED scheme: word 1 = the call's generation, word 2 = 1, word 3 = (generation << 8) | flags,
// valid only if its generation is word 1's.  nerf_hip_read_status decodes whichever scheme word 2 names (the legacy kernels zero it).
constexpr int STATUS_GEN_WORD = 1, STATUS_SCHEME_WORD = 2, STATUS_STAMPED_WORD = 3;
constexpr int STATUS_STICKY_WORD = 32;  // status region = 64 u32 words: [0, 32) cleared by every forward, [32, 64) only by the caller (nerf_hip_read_status_sticky)
constexpr int DBG_WORDS = 16384;   // u64 words of the workspace's diagnostic area (make stamps)
constexpr int DUMP_ROWS = 64;   // extra rows behind every saved tensor / gradient buffer: lanes past the end of a pass store there, unpredicated
constexpr int RAYF = 24;        // floats per ray record

// Ray record (one per ray, written by k_rays):
//  [0..8] R row-major  [9..11] o  [12..14] d_cam  [15..17] d_wrd  [18] near  [19] far
//  [20] (far-near)/(Nc-1)  [21] (far-near)/Nc  [22..23] pad
enum { RF_R = 0, RF_O = 9, RF_DCAM = 12, RF_DWRD = 15, RF_NEAR = 18, RF_FAR = 19, RF_STEP = 20, RF_DELTA = 21 };

// ---- packed weight segments (MFMA fragment order, see pack kernel in field_fwd.hip) ----
// A segment holds W[nft*32 features][KB*8 inputs] as float4[nft][KB][64 lanes]:
//   lane l, component s  =  W[ft*32 + (l&31)][kb*8 + 4*(l>>5) + s]     (0 beyond the real K)
struct Seg { int off4; int nft; int kb; };  // off4 in float4 units
// forward segments.  point_info has no activation (nerf.py:117), so point_info and the feature columns of dir_info are ONE linear
// map of h7: pre_dir = W_dir[:, :24] gamma_d + (W_dir[:, 24:] W_pi) h7 + (W_dir[:, 24:] b_pi + b_dir).  The register kernels run the
// folded 128 x 256 matrix SEG_FOLD (built per call by k_pack_weights, fp32 fma chains) instead of SEG_PI + SEG_DIR: 1,032 of the
// 9,288 MFMAs of a tile less, forward and backward; the LDS-tile kernels (A/B reference) keep the two-step form.
// (the order L5, L6, L7, PI and T_DIR, T_PI, T_L7 ... is relied upon by the tile kernels: consecutive segments of equal size)
constexpr int SEG_L0 = 0, SEG_L1 = 1, SEG_L2 = 2, SEG_L3 = 3, SEG_L4A = 4, SEG_L4B = 5, SEG_L5 = 6, SEG_L6 = 7,
              SEG_L7 = 8, SEG_PI = 9, SEG_DIR = 10, SEG_FOLD = 11, NSEG_FWD = 12;
// transposed segments for the backward dX chain (features <-> inputs swapped)
constexpr int SEG_T_DIR = 12, SEG_T_PI = 13, SEG_T_L7 = 14, SEG_T_L6 = 15, SEG_T_L5 = 16, SEG_T_L4A = 17, SEG_T_L4B = 18,
              SEG_T_L3 = 19, SEG_T_L2 = 20, SEG_T_L1 = 21, SEG_T_L0 = 22, SEG_T_FOLD = 23, NSEG = 24;

__host__ __device__ constexpr int seg_nft(int s) {
  return (s == SEG_DIR || s == SEG_FOLD) ? 4 : (s == SEG_T_L4B || s == SEG_T_L0) ? 2 : 8;
}
__host__ __device__ constexpr int seg_kb(int s) {
  return (s == SEG_L0 || s == SEG_L4B) ? 8 : (s == SEG_T_DIR || s == SEG_T_FOLD) ? 16 : 32;
}
__host__ __device__ constexpr int seg_off4(int s) {
  int o = 0;
  for (int i = 0; i < s; ++i) o += seg_nft(i) * seg_kb(i) * 64;
  return o;
}
constexpr int PACKED_FWD_F4 = seg_off4(NSEG_FWD);
constexpr int PACKED_ALL_F4 = seg_off4(NSEG);

// positional-encoding frequencies f_l = fp32(2^e_l)*fp32(pi), e = linspace(0,L,L) (nerf.py:141-146, quirk Q3);
// bit patterns as produced by torch 2.10 (tests/golden/make_golden.py prints them; tests/test_oracle_golden.py pins them).
__device__ __constant__ const uint32_t kFreqPointBits[10] = {0x40490fdbu, 0x40d928aeu, 0x416a8b6cu, 0x41fd527bu, 0x4288cd33u,
                                                             0x4313c0fau, 0x439f953cu, 0x442c5befu, 0x44ba2881u, 0x45490fdbu};
__device__ __constant__ const uint32_t kFreqDirBits[4] = {0x40490fdbu, 0x40fd527au, 0x419f953cu, 0x42490fdbu};

// weights24 indices
enum { W_L0 = 0, B_L0 = 1, W_SIGMA = 16, B_SIGMA = 17, W_PI = 18, B_PI = 19, W_DIR = 20, B_DIR = 21, W_COLOR = 22, B_COLOR = 23 };

struct Weights24 { const float* p[24]; };
// fp32 scratch of the fold in the workspace: b_fold[128] = W_dir[:, 24:] b_pi (k_pack_weights / k_fold_weights; k_rays adds it to every
// ray's dir_info start vector), then -- bf16-MLP variant only, k_fold_weights -- W_fold[128][256] for the bf16 packers to round
constexpr int FOLD_FLOATS = HALF + HALF * WIDTH;
struct Grads24 { float* p[24]; };

// ---- workspace carve-up (host side, api.cpp) ----
struct WsLayout {
  size_t status, dbg, packed, packed_bf, packed_sp, fold, rayf, dvec, t_c, sig_c, rgb_c, w_c, t_f, sig_f, rgb_f;
  // training-only
  size_t packed_bf_bwd, bsave, bmask, bG, bslabs;  // bf16-MLP training (fragment layout, bf16_common.h)
  size_t packed_sp_bwd, bsave2, bG2;               // split-fp32 training: transposed hi / mid image, the mid-part buffers
  size_t perm, w_m, bundle, save, masks, spre, G, dz, dspre, drgb_c, dsig_c, drgb_f, dsig_f, dt_f, slabs, sbuf, gdbuf, mbuf, dC;
  size_t total;
};

}  // namespace nerf
