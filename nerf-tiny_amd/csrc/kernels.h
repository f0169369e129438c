// kernels.h -- kernel argument blocks and launchers shared by the .hip translation units and api.cpp.
#pragma once
#include "common.h"

#include <atomic>
#include <initializer_list>

namespace nerf {

// More than 64 KiB of dynamic LDS needs an opt-in per kernel -- and per DEVICE: the attribute belongs to the function's
// code object on the current device, so a process that moves to a second GPU must set it again.  `mask` = one bit per
// device ordinal already done (a static at the call site); ordinals >= 64 simply set it on every launch.
inline hipError_t ensure_dynamic_lds(std::atomic<unsigned long long>& mask, std::initializer_list<const void*> kernels, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 64 && ((mask.load(std::memory_order_relaxed) >> dev) & 1ull)) return hipSuccess;
  for (const void* k : kernels) {
    e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  if (dev < 64) mask.fetch_or(1ull << dev, std::memory_order_relaxed);
  return hipSuccess;
}

constexpr int LDA = 260;  // floats per LDS activation row of the field kernels
constexpr int FIELD_LDS_FLOATS = TM * LDA + 4 * TM * 3;

struct FieldArgs {
  const float4* wp;        // packed weights (PACKED_FWD_F4 float4)
  const unsigned char* wbf;  // bf16 weight stream (bf16_common.h), bf16-MLP variant only
  unsigned char* bsave;      // bf16 training: saved layer inputs in fragment layout (bf16_common.h); split-fp32 training: their hi parts
  unsigned char* bsave2;     // split-fp32 training: the mid parts, same layout
  uint16_t* bmask;           // bf16 training: ReLU alive masks
  int wb0, wb_tot;           // first wave block of this pass, wave blocks of both passes
  Weights24 w;             // raw parameter pointers (biases, sigma / colour heads)
  const float* rayf;       // [B][RAYF]
  const float* dvec;       // [B][128]  b_dir + W_dir[:, :24] * gamma_dir(ray)
  const float* t;          // [M] depths
  float* rgb;              // [M][3]
  float* sigma;            // [M]
  float* pts_dbg;          // [M][3] or null
  float* gp_dbg;           // [M][60] or null
  // ---- training only (SAVE): rows of the coarse pass come first, then the fine pass, in every buffer
  float* save;             // [NSAVE][Mtot][256]: h0..h7, c (128 used), gamma_p (64 used)
  uint16_t* masks;         // [8][tiles_tot][4][256]: ReLU masks of h0..h7 in accumulator layout
  float* spre;             // [Mtot] sigma pre-activation
  int row0;                // first row of this pass in the combined buffers
  int tile0;               // first tile of this pass in the mask buffer
  int tiles_tot;
  long long Mtot;
  long long MSrows;        // rows per saved tensor = Mtot + DUMP_ROWS: lanes past the end of a pass store to their own dump row
  int N;                   // samples per ray
  int M;                   // samples of this pass (B*N)
  unsigned long long* stamps;  // diagnostic build (-DNERF_STAMPS) only: [8] cycle sums per phase
};

// saved by the forward: h0..h7, c, gamma_p (feat = point_info's output is not saved: with point_info folded into dir_info no weight
// gradient needs it, see common.h SEG_FOLD)
constexpr int S_H0 = 0, S_C = 8, S_GP = 9, NSAVE = 10;
// gradient buffers written by the backward chain: dpre of layers 0..7 and of dir_info
constexpr int G_L0 = 0, G_D = 8, NGRAD = 9;

struct RaysArgs {
  const int64_t* row;
  const int64_t* col;
  const float* pb;  // [B][17]
  float K[9];
  int B, Nc;
  float* rayf;       // [B][RAYF] or null
  float* dvec;       // [B][128] or null (needs w_dir / b_dir)
  const float* w_dir;  // [128][280]
  const float* b_dir;  // [128]
  const float* b_fold; // [128] W_dir[:, 24:] b_pi (k_fold_weights), added to dvec; or null
  float* t_c;        // [B][Nc] or null
  float* d_cam;      // [B][3] or null
  float* d_wrd;      // [B][3] or null
  unsigned* status;  // the workspace's status words [0, STATUS_STICKY_WORD) are zeroed here (first kernel of a forward), or null
};

struct CoarseArgs {
  const float* t_c;     // [B][Nc]
  const float* sigma;   // [B][Nc]
  const float* rgb;     // [B][Nc][3]
  const float* rayf;    // [B][RAYF] or null (then near_far is used)
  const float* near_far;  // [B][2] or null
  int B, Nc, Nf;
  int delta0_mode;      // 0: from rayf[0] (this batch's ray 0); 1: delta0 given
  float delta0;
  int ray0_override;    // delta0_mode 0 only: 1 = use near0/far0 below instead of rayf[0]
  float near0, far0;
  float* w_c;           // [B][Nc]
  float* C_coarse;      // [B][3]
  float* t_f;           // [B][Nf]
  uint32_t* status;
  uint32_t* sticky;     // or null: a second word that gets the same bits and that no kernel ever clears (nerf_hip_read_status_sticky)
};

struct MergeArgs {
  const float *t_c, *t_f, *sig_c, *sig_f, *rgb_c, *rgb_f;
  int B, Nc, Nf, P;
  float last;
  float* bundle;   // [B][N][5] or null
  float* w;        // [B][N] or null
  uint16_t* perm;  // [B][5][N] or null: sorted position -> original index (coarse first, then fine)
  float* C_fine;   // [B][3]
  int joint;       // NERF_HIP_CORRECTED: ONE stable sort by depth whose permutation moves all five channels (rgb and sigma stay with their
                   // sample) instead of the reference's five independent channel sorts (quirk Q1); perm then holds that permutation five times
};

// SMALL bf16 training batches (Nc = 64, Nf = 128): the per-ray stages ride in the field kernels instead of launches of their own --
// k_coarse / k_merge as EPILOGUES of the coarse / fine forward pass, k_merge_bwd / k_coarse_bwd as PROLOGUES of the fine / coarse chain
// (a workgroup's samples are whole rays: 128 samples = 2 coarse rays or 1 fine ray, 256 = 4 or 2); mode 0 = none
struct FwdFuse {
  int mode;      // 1: coarse composite + resampling behind the coarse pass, 2: merge + sorts + composite behind the fine pass
  CoarseArgs c;
  MergeArgs m;
  // mode 2 inside nerf_hip_train_step: ray_loss's per-element work rides along too (nerf.py:325-331) -- d loss / d C and the summands of the
  // loss, element by element as k_ray_loss forms them; the SUM is taken by the first block of the fine chain launch (BwdFuse), in
  // k_ray_loss's order.  C_true null: no loss here
  const float* C_true;    // [B][3]
  const float* C_coarse;  // [B][3] (written by the coarse pass's epilogue, an earlier launch)
  float *dC_c, *dC_f;     // [B][3]
  float* loss_terms;      // [B][3]
};
hipError_t launch_pack_weights(const Weights24& w, float* fold, float4* out, int nseg, hipStream_t st);  // fold: FOLD_FLOATS scratch (launch_fold_weights runs first)
hipError_t launch_fold_weights(const Weights24& w, float* fold, hipStream_t st);  // fold: FOLD_FLOATS (b_fold, then W_fold): bf16-MLP variant
hipError_t launch_field_fwd(const FieldArgs& a, bool save, hipStream_t st);
hipError_t launch_field_fwd_reg(const FieldArgs& a, bool save, hipStream_t st);
hipError_t launch_field_fwd_bf16(const FieldArgs& a, bool save, hipStream_t st, const FwdFuse* fuse = nullptr);
// the bf16 packers read the fp32 fold (launch_fold_weights, same stream, before them)
hipError_t launch_pack_weights_bf16(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
hipError_t launch_pack_bias_block_bf16(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
hipError_t launch_field_fwd_bf16x(const FieldArgs& a, hipStream_t st);  // inference, 16x16x32 MFMA form
// small batches, Nc = 64 / Nf = 128: the whole inference forward of a ray PAIR per workgroup in one launch (field_fwd_bf16x.hip)
struct PairArgs {
  const unsigned char* wbf;   // 16x16x32 weight image
  RaysArgs rays;              // row, col, pb, K (Nc = 64): the ray records are made in the kernel; rays.rayf / rays.t_c (may be null) get copies
  int B;
  unsigned gen;               // this call's generation (24 bits, never 0): stamps the status flags (common.h STATUS_*)
  int ray0_override;          // quirk Q6: 1 = near0 / far0 below are the batch's GLOBAL ray 0's, 0 = this call's ray 0
  float near0, far0;
  float last;                 // nerf.py:286
  float* C_coarse;            // [B][3]
  float* C_fine;              // [B][3]
  uint32_t* status;           // the workspace's status words (word 0 of the region)
  uint32_t* sticky;
  float *sig_c, *rgb_c, *w_c, *t_f, *sig_f, *rgb_f;  // the workspace's per-sample buffers (written for introspection; may be null)
};
hipError_t launch_render_pair_bf16x(const PairArgs& a, hipStream_t st);
hipError_t launch_pack_weights_bf16x(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
// split-fp32 inference (field_fwd_split.hip): fp32 operands as two bf16 parts, three bf16 MFMAs per product
size_t split_image_bytes();
// split-fp32 training (field_bwd_split.hip): the transposed image of the backward chain, hi and mid fragment of every step interleaved
size_t split_bwd_image_bytes();
hipError_t launch_pack_weights_split_bwd(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
hipError_t launch_pack_weights_split(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
hipError_t launch_field_fwd_split(const FieldArgs& a, bool save, hipStream_t st);  // save: + hi / mid fragment-layout saves, masks, spre (FieldArgs bsave / bsave2 / bmask)
hipError_t launch_rays(const RaysArgs& a, hipStream_t st);
// bf16 paths: fold + packed image(s) + ray records in ONE launch (prep_bf16.hip).  img_fwd: forward image (fwd_form 0: 32x32x16 stream,
// 1: 16x16x32) or null = weights unchanged; img_bwd: transposed image of the backward chain or null; ready: PREP_READY_WORDS u32 words of the
// workspace that no other kernel writes; token: unique per call; sticky: the sticky status word (timeout report) or null
constexpr int PREP_READY_WORDS = 256;
hipError_t launch_prep_bf16(const Weights24& w, float* fold, unsigned char* img_fwd, int fwd_form, unsigned char* img_bwd,
                            unsigned* ready, unsigned token, unsigned* sticky, const RaysArgs& rays, hipStream_t st);
hipError_t launch_coarse(const CoarseArgs& a, hipStream_t st);
size_t merge_lds_bytes(int P);
hipError_t launch_merge(const MergeArgs& a, hipStream_t st);
hipError_t launch_ray_loss(const float* Cc, const float* Cf, const float* Ct, int B, float* loss, float* dCc, float* dCf, hipStream_t st);

}  // namespace nerf

// ================================================================================================
// backward
// ================================================================================================
namespace nerf {

struct FieldBwdArgs {
  const float4* wp;        // packed weights incl. transposed segments
  Weights24 w;
  const float* rayf;       // [B][RAYF]
  const float* t;          // [M] depths of this pass
  const float* rgb;        // [M][3] forward output of this pass
  const float* drgb;       // [M][3] upstream gradient
  const float* dsig;       // [M]    upstream gradient
  const float* save;       // [NSAVE][Mtot][256]
  const uint16_t* masks;   // [8][tiles_tot][4][256]
  const float* spre;       // [Mtot]
  float* G;                // [NGRAD][Mtot][256] pre-activation gradients (inputs of the dW GEMMs)
  float* dz;               // [Mtot][4] colour-head pre-sigmoid gradient
  float* dspre;            // [Mtot]    sigma-head pre-abs gradient
  float* dt;               // [M] fine pass only: in = d loss/d t from the merge, out += direction . d loss/d point
  // bf16-MLP variant (field_bwd_bf16.hip): transposed stream of this pass, ReLU masks, gradient buffer (fragment layout)
  const unsigned char* wbf;
  const uint16_t* bmask;
  unsigned char* bG;
  unsigned char* bG2;      // split-fp32 training (field_bwd_split.hip): the gradients' mid parts (bG holds the hi parts), same layout
  int wb0, wb_tot;
  int row0, tile0, tiles_tot;
  long long Mtot;
  long long MSrows;        // rows per tensor of save / G = Mtot + DUMP_ROWS (see FieldArgs)
  int N, M;
  unsigned long long* stamps;  // diagnostic build only: cycle sums per phase (workspace dbg words [32, 64))
};

// One weight-gradient product dW = G^T X of the fp32 train step (dw_f32.hip).
struct DwItem {
  const float* G;          // [Mtot][256] pre-activation gradients, columns [0, nout)   (thin: the [Mtot][4] buffer dz_r, dz_g, dz_b, dsigma_pre)
  const float* X;          // [Mtot][256] layer inputs, columns [0, nin)                 (thin: c, 128 columns)
  const float* sig;        // has_sig: dsigma_pre, one float per row with a stride of 4 (column 3 of the [Mtot][4] buffer)
  int nout, nin;           // 256 / 128 output columns, 256 / 64 (padded) input columns
  int nin_real;            // input columns that exist in dW (60 of 64 for gamma_p)
  float* dW; int ldw, col0;  // destination [nout][ldw], columns col0 .. col0 + nin_real      (thin: dW_color[3][128])
  float* db;               // [nout] bias gradient = column sums of G, or null                 (thin: db_color[3])
  float* dW2;              // has_sig: dw_sigma[nin] = sum_m sig[m] * X[m][:]
  float* db2;              // thin only: db_sigma[1]
  int thin;                // 1: the colour head (and both heads' bias gradients) as one thin product
  int has_sig;             // 1: the sigma head rides on this product (X = h7): dW2 from the waves that hold X's columns
  float* raysum;           // dir_info product, rows aligned to rays (dw_ray_duty_ok): the column-sum waves also write the per-ray sums of G,
                           // [2][rays][128] (coarse rows, fine rows) -- the scratch of the gamma_d columns' kernels; else null
  int ray_nc, ray_nf;      // samples per ray of the coarse / fine pass
  int rows_c;              // rows of the coarse pass (= rays * ray_nc), the fine pass's rows follow
  int wg0, nwg;            // workgroups [wg0, wg0 + nwg) of the launch work on this product, each on 1/nwg of the rows
  long long slab_off;      // floats: this product's slabs inside DwBatch::slabs
  unsigned long long* stamps;  // diagnostic build only: per-wave (start, end, xcc, hw id) records
};
constexpr int DW_WGS = 256;  // workgroups of the launch: one per CU, dealt out over the products
constexpr int DW_MAX_ITEMS = 13;
// The seven 256 x 256 products share ONE launch (dw_f32.hip: k_dw4_group), DW_GROUP_WGS workgroups each.  The macros are for A/B builds
// (make variant DEFS=-DNERF_DW_GROUP_MAX_ROWS=0: a launch per product, as up to round 2).
#ifndef NERF_DW_GROUP_WGS
#define NERF_DW_GROUP_WGS 36
#endif
#ifndef NERF_DW_GROUP_MAX_ROWS
#define NERF_DW_GROUP_MAX_ROWS (1ll << 40)
#endif
constexpr int DW_GROUP_WGS = NERF_DW_GROUP_WGS;
constexpr long long DW_GROUP_MAX_ROWS = NERF_DW_GROUP_MAX_ROWS;  // rows of G (rays x (Nc + Nf)) up to which the products are grouped
struct DwBatch {
  DwItem item[DW_MAX_ITEMS];
  int n;
  int grouped;             // > 0: items [0, grouped) are 256 x 256 products of equal nwg that share ONE launch (k_dw4_group)
  const float* slabs;
};

// gradients of point_info and of dir_info's feature columns from the folded product (dw_f32.hip: k_fold_grads)
struct FoldGradArgs {
  const float* M;        // [128][256] sum_m dpre_dir[m] (x) h7[m]
  const float* db_dir;   // [128] (final)
  const float* w_dir;    // [128][280]
  const float* w_pi;     // [256][256]
  const float* b_pi;     // [256]
  float* dW_pi;          // [256][256]
  float* db_pi;          // [256]
  float* dW_dir;         // [128][280], columns 24.. written
};
hipError_t launch_fold_grads(const FoldGradArgs& a, hipStream_t st);

struct MergeBwdArgs {
  const float* dC_f;       // [B][3]
  const float* bundle;     // [B][N][5] sorted channels
  const uint16_t* perm;    // [B][5][N]
  int B, Nc, Nf;
  float last;
  float *drgb_c, *dsig_c;  // [B][Nc][3], [B][Nc]   (written: contribution through the merged composite)
  float *drgb_f, *dsig_f, *dt_f;  // [B][Nf][3], [B][Nf], [B][Nf]
};

struct CoarseBwdArgs {
  const float* dC_c;       // [B][3]
  const float* dt_f;       // [B][Nf] total d loss / d t_fine
  const float *t_c, *sigma, *rgb;  // coarse forward values
  const float* rayf;       // [B][RAYF] or null (then near_far is used)
  const float* near_far;   // [B][2] or null
  int B, Nc, Nf;
  int delta0_mode; float delta0;   // as CoarseArgs
  int ray0_override; float near0, far0;
  float *drgb_c, *dsig_c;  // in: merge contribution, out: total
};

struct BwdFuse {
  int mode;      // 1: merged-composite backward in front of the fine chain, 2: resampling + coarse-composite backward in front of the coarse chain
  MergeBwdArgs m;
  CoarseBwdArgs c;
  const float* loss_terms;  // mode 1, or null: block 0 adds up the 3 B summands of the loss in k_ray_loss's order
  float* loss;              // [1]
};

struct SmallGradArgs {
  const float* save;       // saves base
  const float* G;          // grads base
  const float* dz;         // [Mtot][4]
  const float* dspre;      // [Mtot]
  const float* rayf;
  long long Mtot;
  long long MSrows;        // rows per tensor of save / G (Mtot + DUMP_ROWS)
  int B, Nc, Nf;
  float *dW_color, *db_color, *dw_sigma, *db_sigma, *dW_dir;  // destinations (dW_dir: [128][280], cols 0..23 written)
  float* sbuf;             // [2][B][128] scratch: per-ray sums of dpre_dir over the coarse / the fine rows
  int sums_done;           // 1: sbuf was written by the dir_info product (DwItem::raysum); 0: k_dir_ray_sums does it
  float* gdbuf;            // [B][24]  scratch: gamma_dir per ray
};

hipError_t launch_field_bwd(const FieldBwdArgs& a, bool fine, hipStream_t st);
hipError_t launch_field_bwd_split(const FieldBwdArgs& a, bool fine, hipStream_t st);  // split-fp32 training chain (field_bwd_split.hip)
hipError_t launch_field_bwd_reg(const FieldBwdArgs& a, bool fine, hipStream_t st);
hipError_t launch_field_bwd_bf16(const FieldBwdArgs& a, bool fine, hipStream_t st, const BwdFuse* fuse = nullptr);
hipError_t launch_pack_weights_bf16_bwd(const Weights24& w, const float* fold, unsigned char* img, hipStream_t st);
size_t dw_bf16_slab_floats();
// split-fp32 train step: every operand tensor is a (hi, mid) pair of the same layout -- the mid part of a gradient-type tensor (G, Z) lies
// gdelta bytes behind its hi part, the mid part of an input tensor xdelta bytes; {0, 0} = plain bf16 operands
struct DwBfSplit { long long gdelta = 0, xdelta = 0; };
hipError_t launch_dw_bf16_gemm(const unsigned char* G, int g_ks, const unsigned char* X1, int x1_ks, const unsigned char* X2, int x2_ks,
                               const unsigned char* Z, int wb_tot, float* slabs, int* nslab, hipStream_t st, DwBfSplit sp = DwBfSplit{});
hipError_t launch_dw_bf16_group(const unsigned char* const* Gs, const unsigned char* const* Xs, int n, int wb_tot, float* slabs, int* nslab,
                                hipStream_t st, DwBfSplit sp = DwBfSplit{});
// one product of a multi-product launch (small batches: dw_bf16.hip): inputs G, X1 (+ X2) (+ Z); slabs / nslab are filled in
struct DwBfProd {
  const unsigned char* G; int g_ks;
  const unsigned char* X1; int x1_ks;
  const unsigned char* X2; int x2_ks;
  const unsigned char* Z;
  float* slabs; int nslab;
};
// slab_limit: one past the slab space (checked on the host BEFORE the launch: hipErrorOutOfMemory, nothing enqueued) or null
hipError_t launch_dw_bf16_multi(DwBfProd* p, int n, int wb_tot, float* slab_base, const float* slab_limit, float** slab_end, hipStream_t st,
                                DwBfSplit sp = DwBfSplit{});
hipError_t launch_dw_bf16_reduce(const float* slabs, int nslab, int rows, int ni, int o_first, int o_count, int i_first, int i_count,
                                 float* dW, int ldw, int col0, float* db, hipStream_t st);
// dW[o][col0 + i - i_first] = sum over slabs of row o_first + o, column i; db[o] likewise from the last slab column
struct DwBfReduceArgs {
  const float* slabs;
  int nslab, rows, ni;          // slab = [rows][ni + 1]
  int o_first, o_count, i_first, i_count;   // slab columns [i_first, i_first + i_count) -> dW columns col0 ..
  float* dW; int ldw, col0;
  float* db;                    // or null
};
struct DwBfReduceBatch { DwBfReduceArgs r[16]; int n; };
hipError_t launch_dw_bf16_reduce_batch(const DwBfReduceBatch& b, hipStream_t st);  // all slab sums of a step in ONE launch
hipError_t launch_dw(const DwBatch& b, long long Mtot, float* slabs, hipStream_t st, int first = 0, int count = -1);  // products [first, first + count) -> their slabs
hipError_t launch_dw_reduce(const DwBatch& b, hipStream_t st, int first = 0, int count = -1);  // the slabs of items [first, first + count) -> gradients, ONE launch
size_t dw_item_slab_floats(const DwItem& p);
bool dw_ray_duty_ok(const DwItem& p, long long Mtot, int B, int Nc, int Nf);  // may the product of G_dir carry the per-ray sums?
hipError_t launch_small_grads(const SmallGradArgs& a, float* scratch, hipStream_t st);  // scratch: >= 128 * 128 * 24 floats (the slab buffer, after the reduce)
size_t merge_bwd_lds_bytes(int N);
hipError_t launch_merge_bwd(const MergeBwdArgs& a, hipStream_t st);
hipError_t launch_coarse_bwd(const CoarseBwdArgs& a, hipStream_t st);

struct AdamArgs {
  float* param[24];
  const float* grad[24];
  int numel[24];
  int offset[24];          // start of tensor t inside the flat moment buffers
  float* m;                // exp_avg, flat [593,924]
  float* v;                // exp_avg_sq
  float step_size;         // lr / (1 - beta1^t)
  float bias2_sqrt;        // sqrt(1 - beta2^t)
  float one_minus_beta1, beta2, one_minus_beta2, eps;
};

struct GatherArgs {
  const long long* index;  // [B] flat pixel indices
  const float* pixels;     // [n_pic*H*W][3]
  const float* poses;      // [n_pic][17] f32
  int B, H, W;
  long long *row, *col, *pic;  // [B] i64 (the dtypes the reference's collate produces)
  float* pix_val;          // [B][3]
  float* poses_bound;      // [B][17] f32
};

hipError_t launch_adam(const AdamArgs& a, hipStream_t st);
hipError_t launch_gather_rays(const GatherArgs& a, hipStream_t st);

}  // namespace nerf
