/*
 * nerf_hip.h -- C ABI of libnerf_hip.so, the MI355X (gfx950) volume-rendering hot path.
 *
 * The reference (D-Hank/NeRF-tiny) has no FFI: its boundary for this path is the Python
 * method surface of NeRFModel (nerf.py:170 ctor, nerf.py:333-348 forward, nerf.py:325-331
 * ray_loss, autograd backward at nerf.py:473).  Each entry point below replaces the cited
 * span of nerf.py; INTEGRATION.md shows the ctypes stub a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / HIP C++ types.  `stream` is a hipStream_t passed
 *     as void* (NULL = default stream).  Every call only ENQUEUES work on `stream`; no call
 *     synchronises with the host except nerf_hip_read_status() / nerf_hip_read_status_sticky().
 *   - ownership: the caller allocates every buffer (device unless marked HOST) including the
 *     workspace; the library allocates nothing persistent and frees nothing.
 *   - errors: 0 = NERF_HIP_OK, negative = error; nerf_hip_last_error() gives the text
 *     (thread-local).  The library never aborts the process.
 *   - threading: re-entrant per (device, stream); calls sharing a workspace must be ordered
 *     on one stream.
 *   - stream capture: NOT replay-safe.  Every nerf_hip_forward / nerf_hip_train_step bakes a per-call token into
 *     kernel arguments (the in-launch hand-off of the bf16-MLP preparation, the generation stamp of the status word);
 *     a captured graph replayed later would present last replay's token.  Enqueue the calls afresh.
 *   - workspace: the caller zeroes its first 256 bytes (the status region) ONCE after allocating it; the library
 *     never clears words 32..63 of that region on its own (sticky flags, nerf_hip_read_status_sticky).
 *   - all floating point data is IEEE fp32 ("f32"), row-major.
 *
 * Weight order (`weights24`, HOST array of 24 DEVICE pointers) = NeRFModel.network.parameters()
 * order, nn.Linear layout [out, in] (nerf.py:85-99):
 *    0..15  point_layer[i].0.{weight,bias}  i = 0..7   W0[256,60] W1-3[256,256] W4[256,316] W5-7[256,256]
 *   16,17   sigma_layer.0.{weight[1,256],bias[1]}
 *   18,19   point_info.{weight[256,256],bias[256]}
 *   20,21   dir_info.0.{weight[128,280],bias[128]}
 *   22,23   color_layer.0.{weight[3,128],bias[3]}
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_HIP_ABI_VERSION 6 /* 2: NERF_HIP_BF16_MLP, nerf_hip_field_bf16; 3: nerf_hip_backward_overlap, NERF_HIP_SPLIT_MLP;
                                  4: nerf_hip_read_status_sticky; 5: nerf_hip_train_step; 6: NERF_HIP_CORRECTED */

enum {
  NERF_HIP_OK = 0,
  NERF_HIP_ERR_ARG = -1,       /* bad shape / null pointer / unsupported size        */
  NERF_HIP_ERR_WORKSPACE = -2, /* workspace too small for (B, Nc, Nf, flags)          */
  NERF_HIP_ERR_DEVICE = -3,    /* a HIP runtime call failed (text in last_error)      */
  NERF_HIP_ERR_ARCH = -4       /* device is not gfx950                                */
};

/* flags */
enum {
  NERF_HIP_SAVE_FOR_BACKWARD = 1 << 0, /* forward keeps activations in the workspace for nerf_hip_backward */
  NERF_HIP_FORCE_TILE_KERNEL = 1 << 1, /* inference: use the LDS-tile field kernel instead of the register-resident one
                                          (same results up to summation order; for A/B measurements and tests).  With
                                          NERF_HIP_BF16_MLP: inference on the 32x32x16 kernel that training uses instead of the
                                          16x16x32 one (bit-identical to the forward half of a training call) */
  NERF_HIP_BF16_MLP = 1 << 2,          /* BASELINE.json cfg3 "bf16 MLP / fp32 composite": the 12 linear layers of the field MLP
                                          run on bf16 MFMA (bf16-rounded weights and layer inputs, fp32 accumulation and
                                          biases); rays, encodings, compositing, resampling and sort stay fp32.  NOT within
                                          1e-4 of the fp32 reference (about 1e-2); off by default */
  NERF_HIP_WEIGHTS_UNCHANGED = 1 << 3, /* nerf_hip_forward only: the caller guarantees that weights24 hold the same values as in the
                                          previous nerf_hip_forward call on this workspace with the same other flags, so the
                                          packed weight image in the workspace is reused instead of rebuilt (rendering loops) */
  NERF_HIP_SPLIT_MLP = 1 << 4,         /* OPT-IN split-fp32 arithmetic: the linear layers run on bf16 MFMA with every fp32 operand split into two
                                          bf16 parts (hi + mid, 16 significant bits) and three MFMAs per product, fp32 accumulation.
                                          Without NERF_HIP_SAVE_FOR_BACKWARD (inference): within the same 1e-4 bar as the exact-fp32 default
                                          (measured 3e-6 / 2e-5 against the reference's outputs), 3x its rate.  WITH it (the split-fp32 TRAIN
                                          step; pass the flag to nerf_hip_backward / nerf_hip_train_step as well): forward, dX chain and
                                          weight-gradient products in that arithmetic, 2.1x the exact step's rate, loss to 1e-5, gradients inside
                                          the bands the exact path is held to.  Off by default: the default keeps exact k-ordered fp32 fma
                                          chains.  Ignored with NERF_HIP_BF16_MLP */
  NERF_HIP_CORRECTED = 1 << 5,         /* OPTIONAL EXTRA, off by default, NOT the reference's results (SURVEY.md 8a "Q": reproduce the quirks by
                                          default, offer a flagged corrected mode): (Q1) the merged samples are sorted ONCE, by depth, stably, and
                                          rgb / sigma move with their sample, instead of nerf.py:307-308's five independent channel sorts; (Q9) t_fine
                                          is treated as detached: no gradient flows through the fine depths (neither through the sample positions
                                          nor through the merged deltas) into the coarse pass, instead of nerf.py:259's attached t_fine.  Every other
                                          quirk (Q2-Q8, Q10-Q12) stays.  Pass the same flag to forward and backward.  Parity of this mode is
                                          UNPINNED (the reference has no such mode): it is tested against the oracle's restatement of the same
                                          two changes only */
};

/* status word bits (nerf_hip_read_status) */
enum {
  NERF_HIP_STATUS_RESAMPLE_INDEX = 1 << 0, /* the condition on which nerf.py:251-253 calls exit(0) (quirk Q7) */
  NERF_HIP_STATUS_PREP_TIMEOUT = 1 << 1,   /* bf16-MLP calls only: a block of the one-launch preparation gave up waiting for the weight fold
                                              of the SAME launch (bounded wait; relies on in-order workgroup dispatch, which gfx950 provides but
                                              HIP does not promise -- NERF_PREP_BF16=0 in the environment selects separate launches instead).
                                              The packed weight image of that call is then POISONED with NaN: its C_coarse / C_fine / loss /
                                              gradients are NaN, never plausible wrong numbers.  Reported by nerf_hip_read_status for that call
                                              and for every later call that reuses the image (NERF_HIP_WEIGHTS_UNCHANGED), and in the sticky
                                              word.  Never seen outside the fault-injection test. */
};

/* Limits of this build: 2 <= B, 2 <= Nc <= 1024, 1 <= Nf <= 1024, Nc + Nf <= 2048. */

int nerf_hip_abi_version(void);
const char* nerf_hip_last_error(void);

/* Bytes of device workspace nerf_hip_forward / nerf_hip_backward need for these sizes. */
int nerf_hip_ws_bytes(int B, int Nc, int Nf, int flags, size_t* bytes);

/* Byte offset of a named intermediate inside the workspace (introspection for tests / debugging):
 * "t_c" "sig_c" "rgb_c" "w_c" "t_f" "sig_f" "rgb_f", and with NERF_HIP_SAVE_FOR_BACKWARD also "bundle" "w_m" "perm"
 * "save" "G" "dz" "dspre" "drgb_c" "dsig_c" "drgb_f" "dsig_f" "dt_f" (gradient buffers are valid after backward).
 * "save" and "G" are [tensor][B*(Nc+Nf) + 64][256] f32: 64 dump rows follow the real rows of every tensor; "dz" is
 * [B*(Nc+Nf)][4] = (dz_r, dz_g, dz_b, dsigma_pre).  "sbuf" [B][128] = per-ray sums of the dir_info pre-activation gradient,
 * "gdbuf" [B][24] = the direction encodings (both valid after a fp32 backward).  "dbg": 16384 u64 words written only by
 * diagnostic builds (make stamps). */
int nerf_hip_ws_offset(int B, int Nc, int Nf, int flags, const char* name, size_t* offset);

/*
 * Whole forward: replaces NeRFModel.forward -> render_rays (nerf.py:333-348, 286-323).
 *   row, col        [B] i64   pixel coordinates; x <- row, y <- col (nerf.py:186-188)
 *   poses_bound     [B,17] f32  rows as loader.py:33: 3x5 [R|o|hwf] then near, far (already cast, nerf.py:338)
 *   K_inv9          HOST [9] f32, the matrix passed to forward (already transposed, nerf.py:433)
 *   ray0_near_far   HOST [2] f32 or NULL.  The slope of the inverse CDF uses the coarse spacing of
 *                   RAY 0 OF THE BATCH for every ray (nerf.py:233); a caller that shards one batch
 *                   over several GPUs passes the global ray 0's (near, far) here; NULL = this call's ray 0.
 *   last_delta      nerf.py:286 `last` (1e-4)
 *   C_coarse,C_fine [B,3] f32 out
 *   ws              workspace of >= nerf_hip_ws_bytes(B,Nc,Nf,flags) bytes, 256-byte aligned
 */
int nerf_hip_forward(const float* const* weights24, const int64_t* row, const int64_t* col,
                     const float* poses_bound, const float* K_inv9, const float* ray0_near_far,
                     int B, int Nc, int Nf, float last_delta, float* C_coarse, float* C_fine,
                     void* ws, size_t ws_bytes, int flags, void* stream);

/*
 * Backward of nerf_hip_forward (autograd through nerf.py:286-323, called at nerf.py:473).
 * Needs the workspace of a forward run with NERF_HIP_SAVE_FOR_BACKWARD and the same sizes/inputs.
 *   dC_coarse, dC_fine [B,3] f32   upstream gradients
 *   ray0_near_far      as in nerf_hip_forward (must match the forward call)
 *   dweights24         HOST array of 24 DEVICE pointers, same shapes as weights24 (16-byte aligned);
 *                      OVERWRITTEN with the gradients of this batch (sum over rays, no averaging)
 */
int nerf_hip_backward(const float* const* weights24, const float* dC_coarse, const float* dC_fine,
                      const float* ray0_near_far, int B, int Nc, int Nf, float last_delta,
                      float* const* dweights24, void* ws, size_t ws_bytes, int flags, void* stream);

/*
 * nerf_hip_backward for a data-parallel trainer that overlaps its gradient all-reduce with the rest of the backward pass
 * (the collective sits where the reference has loss.backward(); optimizer.step(), nerf.py:473-474): the gradients of
 * point_layer[0..7] (tensors 0..15 of dweights24: 491,520 of the 593,924 parameters) are FINAL at the point where
 * `early_event` (a hipEvent_t created by the caller) is recorded on `stream`; the remaining weight-gradient products (sigma
 * head, point_info, dir_info, colour head: tensors 16..23) follow it.  A caller puts the all-reduce of tensors 0..15 on another
 * stream behind that event and the all-reduce of tensors 16..23 behind the call.  early_event == NULL: nerf_hip_backward.
 */
int nerf_hip_backward_overlap(const float* const* weights24, const float* dC_coarse, const float* dC_fine,
                              const float* ray0_near_far, int B, int Nc, int Nf, float last_delta,
                              float* const* dweights24, void* ws, size_t ws_bytes, int flags, void* stream,
                              void* early_event);

/*
 * One train step's device work in ONE call: nerf_hip_forward (with NERF_HIP_SAVE_FOR_BACKWARD), nerf_hip_ray_loss and
 * nerf_hip_backward_overlap enqueued back to back -- the three calls the reference's loop makes at nerf.py:470-473
 * (`model(...)`, `ray_loss`, `loss.backward()`), without the caller's interpreter between them (at a 400 / 512-ray batch the gaps
 * between three separate calls are 3 % of the step).  Same kernels, same results as the three calls.
 *   C_true            [B,3] f32   the batch's pixel colours
 *   C_coarse, C_fine  [B,3] f32 out (may not be NULL)
 *   loss              [1] f32 out
 *   dweights24, early_event   as in nerf_hip_backward_overlap
 *   flags             NERF_HIP_SAVE_FOR_BACKWARD is implied; ws sized for the flags INCLUDING it
 */
int nerf_hip_train_step(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                        const float* K_inv9, const float* ray0_near_far, const float* C_true, int B, int Nc, int Nf, float last_delta,
                        float* C_coarse, float* C_fine, float* loss, float* const* dweights24, void* ws, size_t ws_bytes, int flags,
                        void* stream, void* early_event);

/* ray_loss (nerf.py:325-331) and its gradient: loss[1] = sum (C_c-C*)^2 + sum (C_f-C*)^2,
 * dC_c = 2 (C_c - C*), dC_f = 2 (C_f - C*).  dC_* may be NULL. */
int nerf_hip_ray_loss(const float* C_coarse, const float* C_fine, const float* C_true, int B,
                      float* loss, float* dC_coarse, float* dC_fine, void* stream);

/* Copies the status word of the last forward on this workspace to the host (synchronises `stream`). */
int nerf_hip_read_status(const void* ws, size_t ws_bytes, uint32_t* status, void* stream);

/* The STICKY status word: every forward ORs the bits of its status word into it as well and no library call clears it, so a train loop
 * that looks at the status only every k-th iteration (one host sync) still sees a condition any iteration in between met -- the reference
 * checks its resampling indices in EVERY forward (nerf.py:251-253).  The caller zeroes the first 256 bytes of a fresh workspace once
 * (the library allocates nothing and cannot know a fresh workspace from a used one); clear != 0 resets the word after reading it.
 * Synchronises `stream`. */
int nerf_hip_read_status_sticky(void* ws, size_t ws_bytes, uint32_t* status, int clear, void* stream);

/*
 * Optional per-kernel timing with HIP events recorded on the caller's stream around every kernel the
 * library launches (used by bench.py for the roofline figure; off by default).  ONE session per process:
 * begin() creates 2*max_launches events; end() waits for them, adds each launch's elapsed ms to
 * ms_sum[kernel id] / count[kernel id] (arrays of n_kernels) and destroys the events.  While a session is open,
 * forward/backward calls from any thread or stream are recorded (slots are handed out under a mutex); begin() and
 * end() themselves must be called from one thread, and end() only after every call of the session has returned.
 */
enum {
  NERF_HIP_K_PACK = 0, NERF_HIP_K_RAYS = 1, NERF_HIP_K_FIELD_COARSE = 2, NERF_HIP_K_COARSE = 3,
  NERF_HIP_K_FIELD_FINE = 4, NERF_HIP_K_MERGE = 5,
  NERF_HIP_K_BWD_MERGE = 6, NERF_HIP_K_BWD_FIELD_FINE = 7, NERF_HIP_K_BWD_COARSE = 8,
  NERF_HIP_K_BWD_FIELD_COARSE = 9, NERF_HIP_K_BWD_DW = 10,
  NERF_HIP_K_RENDER_PAIR = 11, /* small bf16-MLP inference batches: both field passes and both composites of a ray pair in ONE launch */
  NERF_HIP_K_COUNT = 12
};
int nerf_hip_profile_begin(int max_launches);
int nerf_hip_profile_end(double* ms_sum, int* count, int n_kernels);

/* ---------------------------------------------------------------------------------------------
 * Rows f1/f2 of the scope table: what the caller does around the hot path once the renderer is fast.
 * ------------------------------------------------------------------------------------------- */

/* Fused Adam update of all 24 parameter tensors in one launch: torch.optim.Adam semantics as constructed at
 * nerf.py:425 (no weight decay, no amsgrad).  exp_avg / exp_avg_sq: flat [593,924] f32 moment buffers in
 * parameters() order (caller allocated, zero before step 1).  step = 1, 2, ...; lr = this step's learning rate
 * (the LambdaLR of nerf.py:426 is evaluated by the caller). */
int nerf_hip_adam_step(float* const* params24, const float* const* grads24, float* exp_avg, float* exp_avg_sq,
                       int step, float lr, float beta1, float beta2, float eps, void* stream);

/* GPU-resident replacement of NeRFDataset.__getitem__ + DataLoader collation (loader.py:119-133): for B flat pixel
 * indices into pixels[n_pic*H*W][3] (loader.py:88) and poses17[n_pic][17] (f32) writes row, col, pic [B] i64,
 * pix_val [B,3] and poses_bound [B,17] f32 -- the tuple nerf.py:458 iterates over, already on the device. */
int nerf_hip_gather_rays(const int64_t* index, const float* pixels, const float* poses17, int B, int H, int W,
                         int64_t* row, int64_t* col, int64_t* pic, float* pix_val, float* poses_bound, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage entry points (same kernels as nerf_hip_forward; exposed so each row of the hot-path
 * table can be parity-checked on its own).
 * ------------------------------------------------------------------------------------------- */

/* Ray generation (nerf.py:52-67, 186-197, 211, 288): per ray d_cam[B,3], d_wrd[B,3], t_coarse[B,Nc]. */
int nerf_hip_rays(const int64_t* row, const int64_t* col, const float* poses_bound, const float* K_inv9,
                  int B, int Nc, float* d_cam, float* d_wrd, float* t_coarse, void* stream);

/* The field query below with the bf16 MLP of NERF_HIP_BF16_MLP (no debug outputs).
 * ws: >= nerf_hip_ws_bytes(B, N, N, NERF_HIP_BF16_MLP). */
int nerf_hip_field_bf16(const float* const* weights24, const int64_t* row, const int64_t* col,
                        const float* poses_bound, const float* K_inv9, const float* t, int B, int N,
                        float* rgb, float* sigma, void* ws, size_t ws_bytes, void* stream);

/* Field query (nerf.py:200-219 + Encoder 135-167 + Network 101-124) at depths t[B,N]:
 * rgb[B,N,3], sigma[B,N]; optional debug outputs pts[B,N,3], gamma_p[B,N,60] (may be NULL).
 * ws: >= nerf_hip_ws_bytes(B, N, N, 0). */
int nerf_hip_field(const float* const* weights24, const int64_t* row, const int64_t* col,
                   const float* poses_bound, const float* K_inv9, const float* t, int B, int N,
                   float* rgb, float* sigma, float* pts, float* gamma_p,
                   void* ws, size_t ws_bytes, void* stream);

/* Coarse weights, colour and inverse-CDF resampling (nerf.py:263-281, 293-295, 225-261, 320):
 * in t_c, sigma_c [B,Nc], rgb_c [B,Nc,3], near_far [B,2]; out w_c [B,Nc], C_coarse [B,3], t_f [B,Nf],
 * status[1] u32 (OR-ed).  delta0 = coarse spacing used in the slope (HOST value). */
int nerf_hip_coarse_composite(const float* t_c, const float* sigma_c, const float* rgb_c, const float* near_far,
                              float delta0, int B, int Nc, int Nf, float* w_c, float* C_coarse, float* t_f,
                              uint32_t* status, void* stream);

/* Backward of nerf_hip_coarse_composite (autograd through nerf.py:225-281): given dC_coarse[B,3] and the TOTAL
 * d loss/d t_f [B,Nf], ADDS the gradients through C_coarse and through the inverse-CDF resampling to
 * dsig_c[B,Nc] and drgb_c[B,Nc,3] (which already hold the contribution of the merged composite). */
int nerf_hip_coarse_composite_backward(const float* t_c, const float* sigma_c, const float* rgb_c, const float* near_far,
                                       float delta0, int B, int Nc, int Nf, const float* dC_coarse, const float* dt_f,
                                       float* dsig_c, float* drgb_c, void* stream);

/* Merge + per-channel sort + composite (nerf.py:302-321): in t_c/t_f, sigma_c/sigma_f, rgb_c/rgb_f;
 * out sorted bundle[B,Nc+Nf,5] (t,r,g,b,sigma; may be NULL), w[B,Nc+Nf] (may be NULL), C_fine[B,3]. */
int nerf_hip_merge_composite(const float* t_c, const float* t_f, const float* sigma_c, const float* sigma_f,
                             const float* rgb_c, const float* rgb_f, int B, int Nc, int Nf, float last_delta,
                             float* bundle, float* w, float* C_fine, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
