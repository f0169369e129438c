// api.hip -- the extern "C" surface of libnerf_hip.so (include/nerf_hip.h).  Host code only: argument
// checks, workspace carve-up and kernel sequencing on the caller's stream.  No allocation, no host sync
// (except nerf_hip_read_status).
#include "../../include/nerf_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "bf16_common.h"
#include "kernels.h"

using namespace nerf;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(NERF_HIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

int check_sizes(int B, int Nc, int Nf) {
  if (B < 2) return fail(NERF_HIP_ERR_ARG, "B=%d: the reference needs B >= 2 (nerf.py:208 .squeeze())", B);
  if (Nc < 2 || Nc > 1024 || Nf < 1 || Nf > 1024) return fail(NERF_HIP_ERR_ARG, "Nc=%d Nf=%d outside 2..1024 / 1..1024", Nc, Nf);
  // the kernels index samples (rows of the saved tensors, tiles, wave blocks) with 32-bit integers
  if ((long long)B * (Nc + Nf) + DUMP_ROWS + 256 >= (1ll << 31))
    return fail(NERF_HIP_ERR_ARG, "B=%d x (Nc + Nf = %d) samples: a batch must stay below 2^31 samples", B, Nc + Nf);
  return NERF_HIP_OK;
}

// wave blocks (32 samples) of one pass of the bf16 kernels: whole 256-sample workgroups
size_t wave_blocks(int B, int N) { return (((size_t)B * N + 255) / 256) * 8; }

// bf16 weight-gradient phase: all products in one or two launches (dw_bf16.hip: k_dw_bf16_multi) up to this many wave blocks -- where the
// per-product launches' slabs (219 MB per step whatever the batch) and launch boundaries weigh -- and a launch per product beyond.
// NERF_DW_BF16_MULTI=0 / 1 overrides the choice (A/B measurements only).
// NERF_PREP_BF16=0: the bf16 paths prepare with separate launches (fold, pack, rays; the backward packs its own image) as up to round 3
// (A/B measurements only)
bool prep_bf16_disabled() {
  static const bool off = [] { const char* e = getenv("NERF_PREP_BF16"); return e && atoi(e) == 0; }();
  return off;
}
// bf16-MLP inference: the ray-pair kernel up to this many rays (one workgroup per pair = ONE round over the 256 CUs; measured: 400 rays
// -9 %, 512 -6..-9 %, 1024 rays -- a second, half-empty round whose coarse phases run 16-sample waves -- +17 %).
// NERF_PAIR_BF16=0 / 1 overrides the choice (A/B measurements only)
constexpr int PAIR_BF16_MAX_RAYS = 512;
bool pair_bf16(int B) {
  static const int forced = [] { const char* e = getenv("NERF_PAIR_BF16"); return e ? atoi(e) : -1; }();
  return forced >= 0 ? forced != 0 : B <= PAIR_BF16_MAX_RAYS;
}
// large batches: the small-block products of the bf16 weight-gradient phase in one launch (see nerf_hip_backward_overlap).
// NERF_DW_BF16_SMALLGROUP=0 / 1 / 2 selects the variant (A/B measurements); default below
constexpr int DW_BF16_SMALL_GROUP_DEFAULT = 2;  // measured (weight-gradient phase): 2,048 rays 0.767 -> 0.707 ms, 4,096 rays 1.375 -> 1.36; variant 1: +-0
int dw_bf16_small_group() {
  static const int v = [] { const char* e = getenv("NERF_DW_BF16_SMALLGROUP"); return e ? atoi(e) : DW_BF16_SMALL_GROUP_DEFAULT; }();
  return v;
}
// bf16 training: the per-ray stages as epilogues / prologues of the field launches up to this many rays (one round of fine-pass workgroups:
// the stage runs on a quarter of a workgroup's waves).  NERF_FUSE_RAYS=0 / 1 overrides the choice (A/B measurements only)
constexpr int FUSE_RAYS_BF16_MAX_RAYS = 512;
bool fuse_rays_bf16(int B) {
  static const int forced = [] { const char* e = getenv("NERF_FUSE_RAYS"); return e ? atoi(e) : -1; }();
  return forced >= 0 ? forced != 0 : B <= FUSE_RAYS_BF16_MAX_RAYS;
}
constexpr int DW_BF16_MULTI_MAX_WB = 5120;  // 853 rays x (64 + 128); measured: 400 rays -25 %, 512 -18 %, 1024 +-0, 2048 +10 %, 4096 +30 %
bool dw_bf16_multi(int wb_tot) {
  static const int forced = [] { const char* e = getenv("NERF_DW_BF16_MULTI"); return e ? atoi(e) : -1; }();
  return forced >= 0 ? forced != 0 : wb_tot <= DW_BF16_MULTI_MAX_WB;
}


// The eleven weight-gradient products of the fp32 train step (dw_f32.hip) -- ten MFMA-bound ones, one of them carrying the sigma
// head, + the thin colour head -- in launch order, with their slab offsets; pointers are filled in by nerf_hip_backward (null
// here: only sizes matter for the layout).
constexpr int DW_EARLY_ITEMS = 9;  // layers 1..7, layer 4's skip columns, layer 0 = every tensor of point_layer[0..7]
long long build_dw_batch(DwBatch& b, const float* G, const float* save, const float* dz4, size_t MS, float* const* dw, float* mbuf,
                         bool grouped = true) {
  memset(&b, 0, sizeof(b));
  auto add = [&](const float* g, int nout, const float* x, int nin, int nin_real, float* dW, int ldw, int col0, float* db) -> DwItem& {
    DwItem& it = b.item[b.n++];
    it.G = g; it.X = x; it.nout = nout; it.nin = nin; it.nin_real = nin_real; it.dW = dW; it.ldw = ldw; it.col0 = col0; it.db = db;
    return it;
  };
  auto Gt = [&](int t) { return G ? G + (size_t)t * MS : nullptr; };
  auto St = [&](int t) { return save ? save + (size_t)t * MS : nullptr; };
  auto D = [&](int i) { return dw ? dw[i] : nullptr; };
  for (int l = 1; l <= 7; ++l)                                                                   // layers 1..7 (layer 4: hidden columns)
    add(Gt(l), 256, St(l - 1), 256, 256, D(2 * l), l == 4 ? WIDTH + POINT_DIM : WIDTH, 0, D(2 * l + 1));
  add(Gt(4), 256, St(S_GP), 64, POINT_DIM, D(8), WIDTH + POINT_DIM, WIDTH, nullptr);            // layer 4, skip columns
  // layer 0 (X = gamma_p): HBM-bound like the one above; NOT first -- right behind the chain kernel it ran 20 % longer
  add(Gt(0), 256, St(S_GP), 64, POINT_DIM, D(0), POINT_DIM, 0, D(1));
  // point_info folded into dir_info (common.h SEG_FOLD): M = dpre_dir^T h7 (128 x 256) instead of the 256 x 256 point_info product and
  // dir_info's feature columns; k_fold_grads turns M into both tensors' gradients.  The sigma head rides on it (same X = h7).
  DwItem& mi = add(Gt(G_D), 128, St(7), 256, 256, mbuf, WIDTH, 0, D(B_DIR));
  mi.has_sig = 1; mi.sig = dz4 ? dz4 + 3 : nullptr; mi.dW2 = D(W_SIGMA);
  DwItem& th = add(dz4, 32, St(S_C), 128, 128, D(W_COLOR), HALF, 0, D(B_COLOR));                 // colour head (X = c) + the bias gradients of both heads
  th.thin = 1; th.db2 = D(B_SIGMA);
  // The seven 256 x 256 products share one launch with DW_GROUP_WGS workgroups each (dw_f32.hip: k_dw4_group); every other product
  // gets ALL DW_WGS workgroups in a launch of its own.
  b.grouped = grouped ? 7 : 0;
  long long off = 0;
  for (int i = 0; i < b.n; ++i) {
    b.item[i].nwg = i < b.grouped ? DW_GROUP_WGS : DW_WGS;
    b.item[i].wg0 = 0;
    b.item[i].slab_off = off; off += (long long)dw_item_slab_floats(b.item[i]);
  }
  return off;  // slab floats of the whole batch
}

size_t dw_batch_slab_floats() {
  DwBatch b;
  return (size_t)build_dw_batch(b, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
}

WsLayout layout(int B, int Nc, int Nf, int flags) {
  flags &= ~NERF_HIP_WEIGHTS_UNCHANGED;  // not a layout property
  WsLayout L;
  memset(&L, 0, sizeof(L));
  const size_t b = (size_t)B, N = (size_t)Nc + Nf;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
  L.status = take(256);
  L.dbg = take(DBG_WORDS * 8);  // diagnostic builds (-DNERF_STAMPS) write cycle stamps here; untouched otherwise
  L.packed = take((size_t)PACKED_ALL_F4 * 16);
  if (flags & NERF_HIP_BF16_MLP) L.packed_bf = take(BF_IMAGE_BYTES);
  if ((flags & NERF_HIP_SPLIT_MLP) && !(flags & NERF_HIP_BF16_MLP)) L.packed_sp = take(split_image_bytes());
  L.fold = take((FOLD_FLOATS + PREP_READY_WORDS) * 4);  // + the "fold row o is out" words of the one-launch preparation (prep_bf16.hip)
  L.rayf = take(b * RAYF * 4);
  L.dvec = take(b * HALF * 4);
  L.t_c = take(b * Nc * 4);
  L.sig_c = take(b * Nc * 4);
  L.rgb_c = take(b * Nc * 12);
  L.w_c = take(b * Nc * 4);
  L.t_f = take(b * Nf * 4);
  L.sig_f = take(b * Nf * 4);
  L.rgb_f = take(b * Nf * 12);
  if (flags & NERF_HIP_SAVE_FOR_BACKWARD) {
    L.perm = take(b * 5 * N * 2);
    L.w_m = take(b * N * 4);
    L.bundle = take(b * N * 5 * 4);
    const size_t Mtot = b * N;
    const size_t tiles = (b * Nc + TM - 1) / TM + (b * Nf + TM - 1) / TM;
    L.spre = take(Mtot * 4);
    if (flags & NERF_HIP_BF16_MLP) {
      const size_t wb = wave_blocks(B, Nc) + wave_blocks(B, Nf);
      L.packed_bf_bwd = take(BB_IMAGE_BYTES);
      L.bsave = take(wb * BS_TOTAL_KS * BF_FRAG_BYTES);
      L.bmask = take(wb * BM_LAYERS * 1024);
      L.bG = take(wb * BG_TOTAL_KS * BF_FRAG_BYTES);
      L.bslabs = take(dw_bf16_slab_floats() * 4);
      L.mbuf = take((size_t)HALF * WIDTH * 4);
    } else if (flags & NERF_HIP_SPLIT_MLP) {
      // split-fp32 training: the bf16 variant's fragment-layout buffers twice (hi parts, mid parts)
      const size_t wb = wave_blocks(B, Nc) + wave_blocks(B, Nf);
      L.packed_sp_bwd = take(split_bwd_image_bytes());
      L.bsave = take(wb * BS_TOTAL_KS * BF_FRAG_BYTES);
      L.bsave2 = take(wb * BS_TOTAL_KS * BF_FRAG_BYTES);
      L.bmask = take(wb * BM_LAYERS * 1024);
      L.bG = take(wb * BG_TOTAL_KS * BF_FRAG_BYTES);
      L.bG2 = take(wb * BG_TOTAL_KS * BF_FRAG_BYTES);
      L.bslabs = take(dw_bf16_slab_floats() * 4);
      L.mbuf = take((size_t)HALF * WIDTH * 4);
    } else {
      L.save = take((size_t)NSAVE * (Mtot + DUMP_ROWS) * WIDTH * 4);  // + dump rows (kernels.h: MSrows)
      L.masks = take((size_t)8 * tiles * 4 * 256 * 2);
      L.G = take((size_t)NGRAD * (Mtot + DUMP_ROWS) * WIDTH * 4);
      L.dz = take(Mtot * 16);
      L.dspre = take(Mtot * 4);
      L.slabs = take(dw_batch_slab_floats() * 4);  // every product of the step keeps its own slabs: ONE reduce launch at the end
      L.sbuf = take(2 * b * HALF * 4);
      L.gdbuf = take(b * DIR_DIM * 4);
      L.mbuf = take((size_t)HALF * WIDTH * 4);
    }
    L.drgb_c = take(b * Nc * 12);
    L.dsig_c = take(b * Nc * 4);
    L.drgb_f = take(b * Nf * 12);
    L.dsig_f = take(b * Nf * 4);
    L.dt_f = take(b * Nf * 4);
    L.dC = take(b * 9 * 4);  // nerf_hip_train_step: d loss / d C_coarse, d loss / d C_fine, the loss's summands ([B][3] each)
  }
  L.total = o;
  return L;
}

int check_device() {
  static thread_local int checked_dev = -1;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev == checked_dev) return NERF_HIP_OK;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(NERF_HIP_ERR_ARCH, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  checked_dev = dev;
  return NERF_HIP_OK;
}

int check_weights(const float* const* w) {
  if (!w) return fail(NERF_HIP_ERR_ARG, "weights24 is null");
  for (int i = 0; i < 24; ++i) {
    if (!w[i]) return fail(NERF_HIP_ERR_ARG, "weights24[%d] is null", i);
    if (((uintptr_t)w[i] & 15) != 0) return fail(NERF_HIP_ERR_ARG, "weights24[%d] is not 16-byte aligned", i);
  }
  return NERF_HIP_OK;
}

Weights24 as_w24(const float* const* w) {
  Weights24 r;
  for (int i = 0; i < 24; ++i) r.p[i] = w[i];
  return r;
}

// ---- optional per-kernel HIP-event timing (bench.py's roofline leg); off by default ----
// One process-wide session (begin .. end); slots are handed out under a mutex so that callers on several threads /
// streams may run while a session is open (each launch records its own event pair on its own stream).  begin/end
// themselves must not race with each other.
struct Prof {
  std::atomic<bool> on{false};
  std::mutex mu;
  int cap = 0, used = 0, pool_used = 0;
  hipEvent_t* ev = nullptr;     // pool: 2 per launch at most
  hipEvent_t* e0 = nullptr;     // per recorded phase: its start / stop event (handles into the pool)
  hipEvent_t* e1 = nullptr;
  int* kid = nullptr;
} g_prof;

// The phases of one API call follow each other without anything launched in between: the stop event of a phase IS the start
// event of the next one (one event record per boundary instead of two -- each record is a bubble of a few microseconds in front
// of the next kernel).
struct ProfChain {
  hipEvent_t last = nullptr;
};

struct ProfScope {
  hipStream_t st;
  hipEvent_t stop = nullptr;
  ProfChain* chain;
  ProfScope(int kernel_id, hipStream_t s, ProfChain* ch = nullptr) : st(s), chain(ch) {
    if (!g_prof.on.load(std::memory_order_acquire)) return;
    hipEvent_t start = nullptr;
    bool record_start = false;
    {
      std::lock_guard<std::mutex> lk(g_prof.mu);
      if (g_prof.on.load(std::memory_order_relaxed) && g_prof.used < g_prof.cap) {
        const int slot = g_prof.used++;
        g_prof.kid[slot] = kernel_id;
        if (chain && chain->last) {
          start = chain->last;
        } else {
          start = g_prof.ev[g_prof.pool_used++];
          record_start = true;
        }
        stop = g_prof.ev[g_prof.pool_used++];
        g_prof.e0[slot] = start;
        g_prof.e1[slot] = stop;
      }
    }
    if (record_start) (void)hipEventRecord(start, st);
  }
  ~ProfScope() {
    if (stop) (void)hipEventRecord(stop, st);
    if (chain) chain->last = stop;  // (null when this phase was not recorded: the next one records its own start)
  }
};

template <class T>
T* at(void* ws, size_t off) {
  return reinterpret_cast<T*>(static_cast<unsigned char*>(ws) + off);
}

}  // namespace

extern "C" {

int nerf_hip_abi_version(void) { return NERF_HIP_ABI_VERSION; }

const char* nerf_hip_last_error(void) { return g_err; }

int nerf_hip_ws_bytes(int B, int Nc, int Nf, int flags, size_t* bytes) {
  if (!bytes) return fail(NERF_HIP_ERR_ARG, "bytes is null");
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  *bytes = layout(B, Nc, Nf, flags).total;
  return NERF_HIP_OK;
}

}  // extern "C"

namespace {
// what nerf_hip_train_step hands to the forward / backward it is made of: with the per-ray stages fused into the field launches (small
// bf16 batches) ray_loss rides along (kernels.h FwdFuse / BwdFuse) and `fused` comes back true -- the caller then launches no k_ray_loss
struct TrainLoss {
  const float* C_true;
  float *dC_c, *dC_f, *terms, *loss;
  bool fused;
};
int forward_impl(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound, const float* K_inv9,
                 const float* ray0_near_far, int B, int Nc, int Nf, float last_delta, float* C_coarse, float* C_fine, void* ws, size_t ws_bytes,
                 int flags, void* stream, TrainLoss* tl);
int backward_impl(const float* const* weights24, const float* dC_coarse, const float* dC_fine, const float* ray0_near_far, int B, int Nc, int Nf,
                  float last_delta, float* const* dweights24, void* ws, size_t ws_bytes, int flags, void* stream, void* early_event,
                  const TrainLoss* tl);
}  // namespace

extern "C" {

int nerf_hip_forward(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                     const float* K_inv9, const float* ray0_near_far, int B, int Nc, int Nf, float last_delta, float* C_coarse,
                     float* C_fine, void* ws, size_t ws_bytes, int flags, void* stream) {
  return forward_impl(weights24, row, col, poses_bound, K_inv9, ray0_near_far, B, Nc, Nf, last_delta, C_coarse, C_fine, ws, ws_bytes, flags, stream,
                      nullptr);
}

}  // extern "C"

namespace {

int forward_impl(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound, const float* K_inv9,
                 const float* ray0_near_far, int B, int Nc, int Nf, float last_delta, float* C_coarse, float* C_fine, void* ws, size_t ws_bytes,
                 int flags, void* stream, TrainLoss* tl) {
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  if (int rc = check_weights(weights24)) return rc;
  if (!row || !col || !poses_bound || !K_inv9 || !C_coarse || !C_fine || !ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (((uintptr_t)ws & 255) != 0) return fail(NERF_HIP_ERR_ARG, "workspace must be 256-byte aligned");
  const WsLayout L = layout(B, Nc, Nf, flags);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  if (int rc = check_device()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool save = (flags & NERF_HIP_SAVE_FOR_BACKWARD) != 0;
  const bool bf16 = (flags & NERF_HIP_BF16_MLP) != 0;
  // bf16 inference runs on the 16x16x32 MFMA form (field_fwd_bf16x.hip); NERF_HIP_FORCE_TILE_KERNEL selects the training form
  const bool bf16x = bf16 && !save && !(flags & NERF_HIP_FORCE_TILE_KERNEL);
  // split-fp32 inference (field_fwd_split.hip): fp32 operands as two bf16 parts, three bf16 MFMAs per product
  const bool split = (flags & NERF_HIP_SPLIT_MLP) && !bf16;
  // (split && save: the opt-in split-fp32 TRAIN step -- forward with hi / mid fragment-layout saves, field_bwd_split.hip, the two-part weight-gradient products)
  const Weights24 w = as_w24(weights24);

  ProfChain pc;  // the phases below follow each other with nothing in between
  // bf16-MLP calls: fold, packed image(s) -- a training call's transposed image for the backward chain included -- and the ray records
  // in ONE launch (prep_bf16.hip) instead of three or four dependent ones
  const bool one_prep = bf16 && !split && !prep_bf16_disabled();
  // SMALL bf16-MLP inference batches at the shipped sample counts: ONE launch renders every ray pair end to end, ray records included
  // (field_fwd_bf16x.hip: k_render_pair_bf16x); with the weight image reused (rendering loops) it is the only launch of the call
  const bool corrected = (flags & NERF_HIP_CORRECTED) != 0;  // joint depth sort (forward); the fused small-batch forms keep the reference's sorts only
  const bool pair = bf16x && Nc == 64 && Nf == 128 && pair_bf16(B) && !corrected;
  if (!(flags & NERF_HIP_WEIGHTS_UNCHANGED) && !one_prep) {
    ProfScope ps(NERF_HIP_K_PACK, st, &pc);
    if (bf16 || split) HIP_TRY(launch_fold_weights(w, at<float>(ws, L.fold), st));  // fp32 W_fold, b_fold for the bf16 / split packers (bf16_common.h)
    if (split) {
      HIP_TRY(launch_pack_weights_split(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_sp), st));
      if (save) HIP_TRY(launch_pack_weights_split_bwd(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_sp_bwd), st));
    } else
    if (bf16 && bf16x) HIP_TRY(launch_pack_weights_bf16x(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_bf), st));
    else if (bf16) HIP_TRY(launch_pack_weights_bf16(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_bf), st));
    else HIP_TRY(launch_pack_weights(w, at<float>(ws, L.fold), at<float4>(ws, L.packed), save ? NSEG : NSEG_FWD, st));
  }

  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nc;
  ra.rayf = at<float>(ws, L.rayf);
  ra.dvec = (bf16 || split) ? nullptr : at<float>(ws, L.dvec);  // (the bf16 kernels run the direction columns as MFMA k-steps: no per-ray start vector)
  ra.w_dir = w.p[W_DIR]; ra.b_dir = w.p[B_DIR];
  ra.b_fold = at<float>(ws, L.fold);
  ra.t_c = at<float>(ws, L.t_c);
  ra.status = at<unsigned>(ws, L.status);  // zeroed by the kernel (the later kernels OR their flags into it)
  static std::atomic<unsigned> g_token{0};
  unsigned token = ++g_token;
  if ((token & 0xffffffu) == 0) token = ++g_token;  // (its low 24 bits stamp the status word of the pair kernel: never 0)
  if (one_prep) {
    const bool pack = !(flags & NERF_HIP_WEIGHTS_UNCHANGED);
    RaysArgs rp = ra;
    if (pair) rp.B = 0;  // the pair kernel makes its own ray records
    if (pack || rp.B > 0) {
      ProfScope ps(pack ? NERF_HIP_K_PACK : NERF_HIP_K_RAYS, st, &pc);
      HIP_TRY(launch_prep_bf16(w, at<float>(ws, L.fold), pack ? at<unsigned char>(ws, L.packed_bf) : nullptr, bf16x ? 1 : 0,
                               pack && save ? at<unsigned char>(ws, L.packed_bf_bwd) : nullptr,
                               reinterpret_cast<unsigned*>(at<float>(ws, L.fold) + FOLD_FLOATS), token,
                               at<unsigned>(ws, L.status) + STATUS_STICKY_WORD, rp, st));
    }
  } else if (!pair) {
    ProfScope ps(NERF_HIP_K_RAYS, st, &pc);
    HIP_TRY(launch_rays(ra, st));
  }

  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.wp = at<float4>(ws, L.packed);
  if (bf16) fa.wbf = at<unsigned char>(ws, L.packed_bf);
  if (split) fa.wbf = at<unsigned char>(ws, L.packed_sp);
  fa.w = w;
  fa.rayf = at<float>(ws, L.rayf);
  fa.dvec = at<float>(ws, L.dvec);
  // coarse pass (nerf.py:289)
  fa.t = at<float>(ws, L.t_c);
  fa.rgb = at<float>(ws, L.rgb_c);
  fa.sigma = at<float>(ws, L.sig_c);
  fa.N = Nc; fa.M = B * Nc;
#ifdef NERF_STAMPS
  fa.stamps = at<unsigned long long>(ws, L.dbg);  // diagnostic build: cycle sums per phase, words [0, 32)
#endif
  const int tiles_c = (B * Nc + TM - 1) / TM, tiles_f = (B * Nf + TM - 1) / TM;
  if (save) {
    fa.spre = at<float>(ws, L.spre);
    fa.row0 = 0; fa.tile0 = 0; fa.tiles_tot = tiles_c + tiles_f; fa.Mtot = (long long)B * (Nc + Nf); fa.MSrows = fa.Mtot + DUMP_ROWS;
    if (bf16 || split) {
      fa.bsave = at<unsigned char>(ws, L.bsave); fa.bmask = at<uint16_t>(ws, L.bmask);
      if (split) fa.bsave2 = at<unsigned char>(ws, L.bsave2);
      fa.wb0 = 0; fa.wb_tot = (int)(wave_blocks(B, Nc) + wave_blocks(B, Nf));
    } else {
      fa.save = at<float>(ws, L.save); fa.masks = at<uint16_t>(ws, L.masks);
    }
  }
  // SMALL bf16-MLP inference batches at the shipped sample counts: ONE launch renders every ray pair end to end (field_fwd_bf16x.hip:
  // k_render_pair_bf16x -- both field passes, coarse composite + resampling, merge + sorts + composite; bit-identical pixels)
  if (pair) {
    PairArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.wbf = at<unsigned char>(ws, L.packed_bf); pa.rays = ra; pa.rays.status = nullptr; pa.B = B; pa.gen = token & 0xffffffu;
    if (ray0_near_far) { pa.ray0_override = 1; pa.near0 = ray0_near_far[0]; pa.far0 = ray0_near_far[1]; }
    pa.last = last_delta; pa.C_coarse = C_coarse; pa.C_fine = C_fine;
    pa.status = at<uint32_t>(ws, L.status); pa.sticky = at<uint32_t>(ws, L.status) + STATUS_STICKY_WORD;
    pa.sig_c = at<float>(ws, L.sig_c); pa.rgb_c = at<float>(ws, L.rgb_c); pa.w_c = at<float>(ws, L.w_c); pa.t_f = at<float>(ws, L.t_f);
    pa.sig_f = at<float>(ws, L.sig_f); pa.rgb_f = at<float>(ws, L.rgb_f);
    ProfScope ps(NERF_HIP_K_RENDER_PAIR, st, &pc);
    HIP_TRY(launch_render_pair_bf16x(pa, st));
    return NERF_HIP_OK;
  }
  // SMALL bf16 TRAINING batches at the shipped sample counts: k_coarse / k_merge ride as epilogues of the field launches (kernels.h FwdFuse)
  const bool fuse_rays = bf16 && save && Nc == 64 && Nf == 128 && fuse_rays_bf16(B) && !corrected;
  FwdFuse ff;
  memset(&ff, 0, sizeof(ff));
  const bool tile_kernel = (flags & NERF_HIP_FORCE_TILE_KERNEL) != 0;
  auto field = [&](const FieldArgs& f) { return split ? launch_field_fwd_split(f, save, st) : bf16x ? launch_field_fwd_bf16x(f, st) : bf16 ? launch_field_fwd_bf16(f, save, st, ff.mode ? &ff : nullptr) : tile_kernel ? launch_field_fwd(f, save, st) : launch_field_fwd_reg(f, save, st); };

  CoarseArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.t_c = at<float>(ws, L.t_c); ca.sigma = at<float>(ws, L.sig_c); ca.rgb = at<float>(ws, L.rgb_c);
  ca.rayf = at<float>(ws, L.rayf);
  ca.B = B; ca.Nc = Nc; ca.Nf = Nf;
  ca.delta0_mode = 0;
  if (ray0_near_far) { ca.ray0_override = 1; ca.near0 = ray0_near_far[0]; ca.far0 = ray0_near_far[1]; }
  ca.w_c = at<float>(ws, L.w_c); ca.C_coarse = C_coarse; ca.t_f = at<float>(ws, L.t_f);
  ca.status = at<uint32_t>(ws, L.status);
  ca.sticky = at<uint32_t>(ws, L.status) + STATUS_STICKY_WORD;
  if (fuse_rays) {
    ff.mode = 1; ff.c = ca;
    { ProfScope ps(NERF_HIP_K_FIELD_COARSE, st, &pc); HIP_TRY(field(fa)); }
    ff.mode = 0;
  } else {
    { ProfScope ps(NERF_HIP_K_FIELD_COARSE, st, &pc); HIP_TRY(field(fa)); }
    { ProfScope ps(NERF_HIP_K_COARSE, st, &pc); HIP_TRY(launch_coarse(ca, st)); }
  }

  // fine pass (nerf.py:299), same network (quirk Q10)
  fa.t = at<float>(ws, L.t_f);
  fa.rgb = at<float>(ws, L.rgb_f);
  fa.sigma = at<float>(ws, L.sig_f);
  fa.N = Nf; fa.M = B * Nf;
  if (save) { fa.row0 = B * Nc; fa.tile0 = tiles_c; fa.wb0 = (int)wave_blocks(B, Nc); }
  MergeArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.t_c = at<float>(ws, L.t_c); ma.t_f = at<float>(ws, L.t_f);
  ma.sig_c = at<float>(ws, L.sig_c); ma.sig_f = at<float>(ws, L.sig_f);
  ma.rgb_c = at<float>(ws, L.rgb_c); ma.rgb_f = at<float>(ws, L.rgb_f);
  ma.B = B; ma.Nc = Nc; ma.Nf = Nf; ma.P = next_pow2(Nc + Nf);
  ma.last = last_delta;
  ma.joint = corrected ? 1 : 0;
  if (save) { ma.bundle = at<float>(ws, L.bundle); ma.w = at<float>(ws, L.w_m); ma.perm = at<uint16_t>(ws, L.perm); }
  ma.C_fine = C_fine;
  if (fuse_rays) {
    ff.mode = 2; ff.m = ma;
    if (tl) {  // inside nerf_hip_train_step: ray_loss's per-element work rides in the same epilogue
      ff.C_true = tl->C_true; ff.C_coarse = C_coarse; ff.dC_c = tl->dC_c; ff.dC_f = tl->dC_f; ff.loss_terms = tl->terms;
      tl->fused = true;
    }
    { ProfScope ps(NERF_HIP_K_FIELD_FINE, st, &pc); HIP_TRY(field(fa)); }
  } else {
    { ProfScope ps(NERF_HIP_K_FIELD_FINE, st, &pc); HIP_TRY(field(fa)); }
    { ProfScope ps(NERF_HIP_K_MERGE, st, &pc); HIP_TRY(launch_merge(ma, st)); }
  }
  return NERF_HIP_OK;
}

}  // namespace

extern "C" {

int nerf_hip_profile_begin(int max_launches) {
  if (g_prof.ev) return fail(NERF_HIP_ERR_ARG, "profile already active");
  if (max_launches < 1 || max_launches > (1 << 20)) return fail(NERF_HIP_ERR_ARG, "bad max_launches");
  g_prof.ev = new hipEvent_t[2 * (size_t)max_launches];
  g_prof.e0 = new hipEvent_t[max_launches];
  g_prof.e1 = new hipEvent_t[max_launches];
  g_prof.kid = new int[max_launches];
  for (int i = 0; i < 2 * max_launches; ++i) HIP_TRY(hipEventCreate(&g_prof.ev[i]));
  g_prof.cap = max_launches;
  g_prof.used = 0;
  g_prof.pool_used = 0;
  g_prof.on.store(true, std::memory_order_release);
  return NERF_HIP_OK;
}

int nerf_hip_profile_end(double* ms_sum, int* count, int n_kernels) {
  if (!g_prof.ev) return fail(NERF_HIP_ERR_ARG, "profile not active");
  {
    std::lock_guard<std::mutex> lk(g_prof.mu);  // no slot is handed out after this point
    g_prof.on.store(false, std::memory_order_release);
  }
  for (int k = 0; k < n_kernels; ++k) { if (ms_sum) ms_sum[k] = 0.0; if (count) count[k] = 0; }
  int rc = NERF_HIP_OK;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(g_prof.e1[i]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof.e0[i], g_prof.e1[i]);
    if (e != hipSuccess) { rc = fail(NERF_HIP_ERR_DEVICE, "profile event: %s", hipGetErrorString(e)); break; }
    const int k = g_prof.kid[i];
    if (k >= 0 && k < n_kernels) { if (ms_sum) ms_sum[k] += ms; if (count) count[k] += 1; }
  }
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  delete[] g_prof.e0;
  delete[] g_prof.e1;
  delete[] g_prof.kid;
  g_prof.ev = nullptr; g_prof.e0 = g_prof.e1 = nullptr; g_prof.kid = nullptr; g_prof.cap = g_prof.used = g_prof.pool_used = 0;
  return rc;
}

int nerf_hip_backward(const float* const* weights24, const float* dC_coarse, const float* dC_fine, const float* ray0_near_far,
                      int B, int Nc, int Nf, float last_delta, float* const* dweights24, void* ws, size_t ws_bytes, int flags,
                      void* stream) {
  return nerf_hip_backward_overlap(weights24, dC_coarse, dC_fine, ray0_near_far, B, Nc, Nf, last_delta, dweights24, ws, ws_bytes, flags, stream, nullptr);
}

int nerf_hip_backward_overlap(const float* const* weights24, const float* dC_coarse, const float* dC_fine, const float* ray0_near_far,
                              int B, int Nc, int Nf, float last_delta, float* const* dweights24, void* ws, size_t ws_bytes, int flags,
                              void* stream, void* early_event) {
  return backward_impl(weights24, dC_coarse, dC_fine, ray0_near_far, B, Nc, Nf, last_delta, dweights24, ws, ws_bytes, flags, stream, early_event,
                       nullptr);
}

}  // extern "C"

namespace {

int backward_impl(const float* const* weights24, const float* dC_coarse, const float* dC_fine, const float* ray0_near_far, int B, int Nc, int Nf,
                  float last_delta, float* const* dweights24, void* ws, size_t ws_bytes, int flags, void* stream, void* early_event,
                  const TrainLoss* tl) {
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  if (int rc = check_weights(weights24)) return rc;
  if (int rc = check_weights(const_cast<const float* const*>(dweights24))) return rc;
  if (!dC_coarse || !dC_fine || !ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (!(flags & NERF_HIP_SAVE_FOR_BACKWARD)) return fail(NERF_HIP_ERR_ARG, "backward needs a forward run with NERF_HIP_SAVE_FOR_BACKWARD");
  const bool bf16 = (flags & NERF_HIP_BF16_MLP) != 0;
  const bool split = (flags & NERF_HIP_SPLIT_MLP) && !bf16;  // split-fp32 train step: field_bwd_split.hip + the two-part weight-gradient products
  const WsLayout L = layout(B, Nc, Nf, flags);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  if (int rc = check_device()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Weights24 w = as_w24(weights24);
  float* const* dw = dweights24;
  const long long Mtot = (long long)B * (Nc + Nf);
  const size_t MS = (size_t)(Mtot + DUMP_ROWS) * WIDTH;  // tensor stride of save / G
  const int tiles_c = (B * Nc + TM - 1) / TM, tiles_f = (B * Nf + TM - 1) / TM;
  float* save = at<float>(ws, L.save);
  float* G = at<float>(ws, L.G);
  const int wb_c = (int)wave_blocks(B, Nc), wb_tot = wb_c + (int)wave_blocks(B, Nf);
  ProfChain pc;  // the phases below follow each other with nothing in between
  // (the fp32 fold of the forward call is still in the workspace: backward runs on the weights its forward ran on)
  // (bf16: the forward call packed the transposed image of the chain too -- prep_bf16.hip -- unless the one-launch preparation is off)
  if (bf16 && prep_bf16_disabled()) { ProfScope ps(NERF_HIP_K_PACK, st, &pc); HIP_TRY(launch_pack_weights_bf16_bwd(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_bf_bwd), st)); }

  // 1. merged composite + per-channel sort backward (nerf.py:302-321)
  MergeBwdArgs mb;
  memset(&mb, 0, sizeof(mb));
  mb.dC_f = dC_fine; mb.bundle = at<float>(ws, L.bundle); mb.perm = at<uint16_t>(ws, L.perm);
  mb.B = B; mb.Nc = Nc; mb.Nf = Nf; mb.last = last_delta;
  mb.drgb_c = at<float>(ws, L.drgb_c); mb.dsig_c = at<float>(ws, L.dsig_c);
  mb.drgb_f = at<float>(ws, L.drgb_f); mb.dsig_f = at<float>(ws, L.dsig_f); mb.dt_f = at<float>(ws, L.dt_f);
  // SMALL bf16 batches: the per-ray backward stages ride as prologues of the chain launches (kernels.h BwdFuse)
  const bool corrected = (flags & NERF_HIP_CORRECTED) != 0;
  const bool fuse_rays = bf16 && Nc == 64 && Nf == 128 && fuse_rays_bf16(B) && !corrected;  // (must agree with the forward's choice)
  BwdFuse bz;
  memset(&bz, 0, sizeof(bz));
  if (!fuse_rays) { ProfScope ps(NERF_HIP_K_BWD_MERGE, st, &pc); HIP_TRY(launch_merge_bwd(mb, st)); }

  // 2. fine-pass field backward (dX chain incl. d loss / d t_fine)
  FieldBwdArgs fb;
  memset(&fb, 0, sizeof(fb));
  fb.wp = at<float4>(ws, L.packed); fb.w = w; fb.rayf = at<float>(ws, L.rayf);
  fb.spre = at<float>(ws, L.spre);
  if (bf16) {
    fb.wbf = at<unsigned char>(ws, L.packed_bf_bwd); fb.bmask = at<uint16_t>(ws, L.bmask); fb.bG = at<unsigned char>(ws, L.bG);
    fb.wb_tot = wb_tot;
  } else if (split) {
    fb.wbf = at<unsigned char>(ws, L.packed_sp_bwd); fb.bmask = at<uint16_t>(ws, L.bmask);
    fb.bG = at<unsigned char>(ws, L.bG); fb.bG2 = at<unsigned char>(ws, L.bG2);
    fb.wb_tot = wb_tot;
  } else {
    fb.save = save; fb.masks = at<uint16_t>(ws, L.masks);
    fb.G = G; fb.dz = at<float>(ws, L.dz); fb.dspre = at<float>(ws, L.dspre);
  }
  fb.tiles_tot = tiles_c + tiles_f; fb.Mtot = Mtot; fb.MSrows = Mtot + DUMP_ROWS;
#ifdef NERF_STAMPS
  fb.stamps = at<unsigned long long>(ws, L.dbg) + 32;
#endif
  fb.t = at<float>(ws, L.t_f); fb.rgb = at<float>(ws, L.rgb_f);
  fb.drgb = at<float>(ws, L.drgb_f); fb.dsig = at<float>(ws, L.dsig_f); fb.dt = at<float>(ws, L.dt_f);
  fb.row0 = B * Nc; fb.tile0 = tiles_c; fb.N = Nf; fb.M = B * Nf; fb.wb0 = wb_c;
  const bool tile_kernel = (flags & NERF_HIP_FORCE_TILE_KERNEL) != 0;
  auto chain = [&](const FieldBwdArgs& f, bool fine) { return split ? launch_field_bwd_split(f, fine, st) : bf16 ? launch_field_bwd_bf16(f, fine, st, bz.mode ? &bz : nullptr) : tile_kernel ? launch_field_bwd(f, fine, st) : launch_field_bwd_reg(f, fine, st); };
  if (fuse_rays) {
    bz.mode = 1; bz.m = mb;
    if (tl && tl->fused) { bz.loss_terms = tl->terms; bz.loss = tl->loss; }  // block 0 of the fine chain adds up the loss's summands
  }
  { ProfScope ps(NERF_HIP_K_BWD_FIELD_FINE, st, &pc); HIP_TRY(chain(fb, true)); }
  bz.mode = 0;
  // NERF_HIP_CORRECTED: t_fine detached -- what the merge backward (deltas) and the fine chain (sample positions) left in d loss / d t_fine is
  // dropped before the resampling backward reads it, so the coarse pass receives the gradient of C_coarse alone
  if (corrected) HIP_TRY(hipMemsetAsync(at<float>(ws, L.dt_f), 0, (size_t)B * Nf * sizeof(float), st));

  // 3. resampling + coarse composite backward (nerf.py:225-261, 263-281)
  CoarseBwdArgs cb;
  memset(&cb, 0, sizeof(cb));
  cb.dC_c = dC_coarse; cb.dt_f = at<float>(ws, L.dt_f);
  cb.t_c = at<float>(ws, L.t_c); cb.sigma = at<float>(ws, L.sig_c); cb.rgb = at<float>(ws, L.rgb_c);
  cb.rayf = at<float>(ws, L.rayf);
  cb.B = B; cb.Nc = Nc; cb.Nf = Nf;
  if (ray0_near_far) { cb.ray0_override = 1; cb.near0 = ray0_near_far[0]; cb.far0 = ray0_near_far[1]; }
  cb.drgb_c = at<float>(ws, L.drgb_c); cb.dsig_c = at<float>(ws, L.dsig_c);
  if (fuse_rays) { bz.mode = 2; bz.c = cb; }
  else { ProfScope ps(NERF_HIP_K_BWD_COARSE, st, &pc); HIP_TRY(launch_coarse_bwd(cb, st)); }

  // 4. coarse-pass field backward
  fb.t = at<float>(ws, L.t_c); fb.rgb = at<float>(ws, L.rgb_c);
  fb.drgb = at<float>(ws, L.drgb_c); fb.dsig = at<float>(ws, L.dsig_c); fb.dt = nullptr;
  fb.row0 = 0; fb.tile0 = 0; fb.N = Nc; fb.M = B * Nc; fb.wb0 = 0;
  { ProfScope ps(NERF_HIP_K_BWD_FIELD_COARSE, st, &pc); HIP_TRY(chain(fb, false)); }

  // 5. weight gradients: dW = G^T X over all B*(Nc+Nf) samples
  // the bf16 products over a pair of fragment-layout buffers (layer inputs `bs`, pre-activation gradients `bg`) into the 24 gradient tensors `dwp`
  // and the folded product's M (`mbuf`); `fold`: finish with k_fold_grads.  `sp` != {0, 0}: the split-fp32 train step -- `bs` / `bg` hold the hi
  // parts, the mid parts lie sp.xdelta / sp.gdelta bytes behind, and every product is formed as hi x hi + hi x mid + mid x hi in ONE pass.
  auto dw_bf16_phase = [&](const unsigned char* bs, const unsigned char* bg, float* const* dwp, float* mbuf, void* ev, bool fold, DwBfSplit sp) -> int {
    float* slabs = at<float>(ws, L.bslabs);
    const float* const slab_limit = slabs + dw_bf16_slab_floats();  // checked by launch_dw_bf16_multi before it enqueues anything
    auto X = [&](int t) { return bs + (size_t)wb_tot * bs_cum(t) * BF_FRAG_BYTES; };
    auto Gt = [&](int t) { return bg + (size_t)wb_tot * bg_cum(t) * BF_FRAG_BYTES; };
    int ns = 0;
    // every product writes its own slab region; ONE launch sums them all at the end
    DwBfReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    auto red = [&](const float* sl, int nslab, int rows, int ni, int o_first, int o_count, int i_first, int i_count, float* dW, int ldw, int col0, float* db) {
      DwBfReduceArgs& r = rb.r[rb.n++];
      r.slabs = sl; r.nslab = nslab; r.rows = rows; r.ni = ni; r.o_first = o_first; r.o_count = o_count; r.i_first = i_first; r.i_count = i_count;
      r.dW = dW; r.ldw = ldw; r.col0 = col0; r.db = db;
    };
    if (dw_bf16_multi(wb_tot)) {
      // SMALL batch (a rank's share of a strong-scaling step, the reference's 400-ray batch): all products of the early part in ONE
      // launch and all of the late part in another (ONE launch for everything without an early event), the workgroups dealt out in
      // proportion to the products' bytes: a quarter of the slab traffic, three or four launch boundaries fewer, long streams
      static const int layers[6] = {1, 2, 3, 5, 6, 7};
      DwBfProd pr[10];
      memset(pr, 0, sizeof(pr));
      int n = 0;
      pr[n++] = DwBfProd{Gt(BG_L0), 16, X(BS_GP), 4, nullptr, 0, nullptr, nullptr, 0};                                   // 0: layer 0 (X = gamma_p)
      for (int k = 0; k < 6; ++k) pr[n++] = DwBfProd{Gt(BG_L0 + layers[k]), 16, X(BS_H0 + layers[k] - 1), 16, nullptr, 0, nullptr, nullptr, 0};  // 1..6
      pr[n++] = DwBfProd{Gt(BG_L0 + 4), 16, X(BS_H0 + 3), 16, X(BS_GP), 4, nullptr, nullptr, 0};                        // 7: layer 4, X = [h3 | gamma_p]
      const int n_early = n;
      pr[n++] = DwBfProd{Gt(BG_D), 8, X(BS_GD), 2, X(BS_H0 + 7), 16, Gt(BG_Z), nullptr, 0};                              // 8: folded dir_info product + sigma head
      pr[n++] = DwBfProd{Gt(BG_Z), 2, X(BS_C), 8, nullptr, 0, nullptr, nullptr, 0};                                       // 9: colour head
      auto early_reds = [&]() {
        red(pr[0].slabs, pr[0].nslab, 256, 64, 0, 256, 0, POINT_DIM, dwp[0], POINT_DIM, 0, dwp[1]);
        for (int k = 0; k < 6; ++k) red(pr[1 + k].slabs, pr[1 + k].nslab, 256, 256, 0, 256, 0, WIDTH, dwp[2 * layers[k]], WIDTH, 0, dwp[2 * layers[k] + 1]);
        red(pr[7].slabs, pr[7].nslab, 256, 320, 0, 256, 0, WIDTH + POINT_DIM, dwp[8], WIDTH + POINT_DIM, 0, dwp[9]);
      };
      auto late_reds = [&]() {
        red(pr[8].slabs, pr[8].nslab, 160, 288, 0, HALF, 0, DIR_DIM, dwp[W_DIR], WIDTH + DIR_DIM, 0, dwp[B_DIR]);
        red(pr[8].slabs, pr[8].nslab, 160, 288, 0, HALF, 32, WIDTH, mbuf, WIDTH, 0, nullptr);
        red(pr[8].slabs, pr[8].nslab, 160, 288, HALF + 3, 1, 32, WIDTH, dwp[W_SIGMA], WIDTH, 0, nullptr);
        red(pr[9].slabs, pr[9].nslab, 32, 128, 0, 3, 0, HALF, dwp[W_COLOR], HALF, 0, dwp[B_COLOR]);
        red(pr[9].slabs, pr[9].nslab, 32, 128, 3, 1, 0, 0, nullptr, 0, 0, dwp[B_SIGMA]);
      };
      float* end = slabs;
      if (ev) {
        HIP_TRY(launch_dw_bf16_multi(pr, n_early, wb_tot, slabs, slab_limit, &end, st, sp));
        early_reds();
        HIP_TRY(launch_dw_bf16_reduce_batch(rb, st));
        HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), st));
        rb.n = 0;
        HIP_TRY(launch_dw_bf16_multi(pr + n_early, n - n_early, wb_tot, end, slab_limit, &end, st, sp));
        late_reds();
      } else {
        HIP_TRY(launch_dw_bf16_multi(pr, n, wb_tot, slabs, slab_limit, &end, st, sp));
        early_reds();
        late_reds();
      }
      HIP_TRY(launch_dw_bf16_reduce_batch(rb, st));
      FoldGradArgs fg;
      fg.M = mbuf; fg.db_dir = dwp[B_DIR]; fg.w_dir = w.p[W_DIR]; fg.w_pi = w.p[W_PI]; fg.b_pi = w.p[B_PI];
      fg.dW_pi = dwp[W_PI]; fg.db_pi = dwp[B_PI]; fg.dW_dir = dwp[W_DIR];
      if (fold) HIP_TRY(launch_fold_grads(fg, st));
      return NERF_HIP_OK;
    }
    // LARGE batches, the products with SMALL blocks (layer 0: 20 KiB per wave block, the folded dir_info product 28, the colour head 10): alone
    // in a launch each is paced by the ring's per-block latency (3.4 / 4.3 / 3.8 TB/s against the 5.8 of the 256 x 256 products:
    // profiles/r03_train_bf16_pmc.json); sharing ONE launch (k_dw_bf16_multi, workgroups dealt out by cost) their streams overlap.
    // Variant 1: six | layer 4 | {layer 0, folded, colour}; 2: six | {layer 4, layer 0, folded, colour}; 0: a launch each (round 3).
    // Without an early event only (the overlap needs layer 0 in front of the event and the other two behind it).
    const int small_group = ev ? 0 : dw_bf16_small_group();
    if (small_group) {
      static const int layers[6] = {1, 2, 3, 5, 6, 7};
      const unsigned char* Gs[6];
      const unsigned char* Xs[6];
      for (int k = 0; k < 6; ++k) { Gs[k] = Gt(BG_L0 + layers[k]); Xs[k] = X(BS_H0 + layers[k] - 1); }
      HIP_TRY(launch_dw_bf16_group(Gs, Xs, 6, wb_tot, slabs, &ns, st, sp));
      for (int k = 0; k < 6; ++k)
        red(slabs + (size_t)k * ns * 256 * 257, ns, 256, 256, 0, 256, 0, WIDTH, dwp[2 * layers[k]], WIDTH, 0, dwp[2 * layers[k] + 1]);
      slabs += (size_t)6 * ns * 256 * 257;
      DwBfProd pr[4];
      memset(pr, 0, sizeof(pr));
      int n = 0, i_l4 = -1;
      if (small_group == 1) {
        HIP_TRY(launch_dw_bf16_gemm(Gt(BG_L0 + 4), 16, X(BS_H0 + 3), 16, X(BS_GP), 4, nullptr, wb_tot, slabs, &ns, st, sp));
        red(slabs, ns, 256, 320, 0, 256, 0, WIDTH + POINT_DIM, dwp[8], WIDTH + POINT_DIM, 0, dwp[9]);
        slabs += (size_t)ns * 256 * 321;
      } else {
        i_l4 = n;
        pr[n++] = DwBfProd{Gt(BG_L0 + 4), 16, X(BS_H0 + 3), 16, X(BS_GP), 4, nullptr, nullptr, 0};
      }
      const int i_l0 = n;
      pr[n++] = DwBfProd{Gt(BG_L0), 16, X(BS_GP), 4, nullptr, 0, nullptr, nullptr, 0};
      const int i_d = n;
      pr[n++] = DwBfProd{Gt(BG_D), 8, X(BS_GD), 2, X(BS_H0 + 7), 16, Gt(BG_Z), nullptr, 0};
      const int i_c = n;
      pr[n++] = DwBfProd{Gt(BG_Z), 2, X(BS_C), 8, nullptr, 0, nullptr, nullptr, 0};
      float* end = slabs;
      HIP_TRY(launch_dw_bf16_multi(pr, n, wb_tot, slabs, slab_limit, &end, st, sp));
      if (i_l4 >= 0) red(pr[i_l4].slabs, pr[i_l4].nslab, 256, 320, 0, 256, 0, WIDTH + POINT_DIM, dwp[8], WIDTH + POINT_DIM, 0, dwp[9]);
      red(pr[i_l0].slabs, pr[i_l0].nslab, 256, 64, 0, 256, 0, POINT_DIM, dwp[0], POINT_DIM, 0, dwp[1]);
      red(pr[i_d].slabs, pr[i_d].nslab, 160, 288, 0, HALF, 0, DIR_DIM, dwp[W_DIR], WIDTH + DIR_DIM, 0, dwp[B_DIR]);
      red(pr[i_d].slabs, pr[i_d].nslab, 160, 288, 0, HALF, 32, WIDTH, mbuf, WIDTH, 0, nullptr);
      red(pr[i_d].slabs, pr[i_d].nslab, 160, 288, HALF + 3, 1, 32, WIDTH, dwp[W_SIGMA], WIDTH, 0, nullptr);
      red(pr[i_c].slabs, pr[i_c].nslab, 32, 128, 0, 3, 0, HALF, dwp[W_COLOR], HALF, 0, dwp[B_COLOR]);
      red(pr[i_c].slabs, pr[i_c].nslab, 32, 128, 3, 1, 0, 0, nullptr, 0, 0, dwp[B_SIGMA]);
      HIP_TRY(launch_dw_bf16_reduce_batch(rb, st));
      FoldGradArgs fg;
      fg.M = mbuf; fg.db_dir = dwp[B_DIR]; fg.w_dir = w.p[W_DIR]; fg.w_pi = w.p[W_PI]; fg.b_pi = w.p[B_PI];
      fg.dW_pi = dwp[W_PI]; fg.db_pi = dwp[B_PI]; fg.dW_dir = dwp[W_DIR];
      if (fold) HIP_TRY(launch_fold_grads(fg, st));
      return NERF_HIP_OK;
    }
    // layer 0: X = gamma_p
    HIP_TRY(launch_dw_bf16_gemm(Gt(BG_L0), 16, X(BS_GP), 4, nullptr, 0, nullptr, wb_tot, slabs, &ns, st, sp));
    red(slabs, ns, 256, 64, 0, 256, 0, POINT_DIM, dwp[0], POINT_DIM, 0, dwp[1]);
    slabs += (size_t)ns * 256 * 65;
    {  // layers 1, 2, 3, 5, 6, 7: six 256 x 256 products in ONE launch (42 workgroups each: a sixth of the slab traffic)
      static const int layers[6] = {1, 2, 3, 5, 6, 7};
      const unsigned char* Gs[6];
      const unsigned char* Xs[6];
      for (int k = 0; k < 6; ++k) { Gs[k] = Gt(BG_L0 + layers[k]); Xs[k] = X(BS_H0 + layers[k] - 1); }
      HIP_TRY(launch_dw_bf16_group(Gs, Xs, 6, wb_tot, slabs, &ns, st, sp));
      for (int k = 0; k < 6; ++k)
        red(slabs + (size_t)k * ns * 256 * 257, ns, 256, 256, 0, 256, 0, WIDTH, dwp[2 * layers[k]], WIDTH, 0, dwp[2 * layers[k] + 1]);
      slabs += (size_t)6 * ns * 256 * 257;
    }
    // layer 4: one pass over dpre4 for both column groups of the [256][316] matrix: X = [h3 | gamma_p]
    HIP_TRY(launch_dw_bf16_gemm(Gt(BG_L0 + 4), 16, X(BS_H0 + 3), 16, X(BS_GP), 4, nullptr, wb_tot, slabs, &ns, st, sp));
    red(slabs, ns, 256, 320, 0, 256, 0, WIDTH + POINT_DIM, dwp[8], WIDTH + POINT_DIM, 0, dwp[9]);
    slabs += (size_t)ns * 256 * 321;
    if (ev) {  // point_layer[0..7] are complete here: their sums go out now, the rest of the products follow the event
      HIP_TRY(launch_dw_bf16_reduce_batch(rb, st));
      HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), st));
      rb.n = 0;
    }
    // point_info folded into dir_info (bf16_common.h): ONE product dpre_dir^T [gamma_d | h7] -- columns 0..23 are dir_info's direction
    // columns, columns 32.. are M = dpre_dir^T h7 (-> k_fold_grads below) -- and in the same pass over h7 the sigma head: row 3 of h7^T (dz, dspre)
    HIP_TRY(launch_dw_bf16_gemm(Gt(BG_D), 8, X(BS_GD), 2, X(BS_H0 + 7), 16, Gt(BG_Z), wb_tot, slabs, &ns, st, sp));
    red(slabs, ns, 160, 288, 0, HALF, 0, DIR_DIM, dwp[W_DIR], WIDTH + DIR_DIM, 0, dwp[B_DIR]);
    red(slabs, ns, 160, 288, 0, HALF, 32, WIDTH, mbuf, WIDTH, 0, nullptr);
    red(slabs, ns, 160, 288, HALF + 3, 1, 32, WIDTH, dwp[W_SIGMA], WIDTH, 0, nullptr);
    slabs += (size_t)ns * 160 * 289;
    // colour head = rows 0..2 of the (dz, dspre) tile against c; row 3 of its column sums = the sigma bias gradient
    HIP_TRY(launch_dw_bf16_gemm(Gt(BG_Z), 2, X(BS_C), 8, nullptr, 0, nullptr, wb_tot, slabs, &ns, st, sp));
    red(slabs, ns, 32, 128, 0, 3, 0, HALF, dwp[W_COLOR], HALF, 0, dwp[B_COLOR]);
    red(slabs, ns, 32, 128, 3, 1, 0, 0, nullptr, 0, 0, dwp[B_SIGMA]);
    HIP_TRY(launch_dw_bf16_reduce_batch(rb, st));
    {
      FoldGradArgs fg;
      fg.M = mbuf; fg.db_dir = dwp[B_DIR]; fg.w_dir = w.p[W_DIR]; fg.w_pi = w.p[W_PI]; fg.b_pi = w.p[B_PI];
      fg.dW_pi = dwp[W_PI]; fg.db_pi = dwp[B_PI]; fg.dW_dir = dwp[W_DIR];
      if (fold) HIP_TRY(launch_fold_grads(fg, st));
    }
    return NERF_HIP_OK;
  };
  if (bf16) {
    ProfScope ps(NERF_HIP_K_BWD_DW, st, &pc);
    if (int rc = dw_bf16_phase(at<unsigned char>(ws, L.bsave), at<unsigned char>(ws, L.bG), dw, at<float>(ws, L.mbuf), early_event, true, DwBfSplit{})) return rc;
  } else if (split) {
    // split-fp32 train step: G = G_hi + G_mid, X = X_hi + X_mid (bf16 parts, fragment layout): G^T X = G_hi^T X_hi + G_hi^T X_mid + G_mid^T X_hi
    // + O(2^-16), formed block by block inside the two-part instantiations of the bf16 products (dw_bf16.hip): ONE pass over the four buffers
    ProfScope ps(NERF_HIP_K_BWD_DW, st, &pc);
    DwBfSplit sp;
    sp.xdelta = (long long)L.bsave2 - (long long)L.bsave;
    sp.gdelta = (long long)L.bG2 - (long long)L.bG;
    if (int rc = dw_bf16_phase(at<unsigned char>(ws, L.bsave), at<unsigned char>(ws, L.bG), dw, at<float>(ws, L.mbuf), early_event, true, sp)) return rc;
  } else {
    ProfScope ps(NERF_HIP_K_BWD_DW, st, &pc);
    DwBatch batch;
    build_dw_batch(batch, G, save, at<float>(ws, L.dz), MS, dw, at<float>(ws, L.mbuf), Mtot <= DW_GROUP_MAX_ROWS);
    float* slabs = at<float>(ws, L.slabs);
    batch.slabs = slabs;
#ifdef NERF_STAMPS
    batch.item[0].stamps = at<unsigned long long>(ws, L.dbg) + 64;  // layer 1: a 256 x 256 product
#endif
    // the dir_info product writes the per-ray sums of its G operand on the way when its row ranges line up with the rays
    bool ray_duty = false;
    for (int i = 0; i < batch.n; ++i) {
      DwItem& p = batch.item[i];
      if (p.G == G + (size_t)G_D * MS && dw_ray_duty_ok(p, Mtot, B, Nc, Nf)) {
        p.raysum = at<float>(ws, L.sbuf); p.ray_nc = Nc; p.ray_nf = Nf; p.rows_c = B * Nc;
        ray_duty = true;
      }
    }
    if (early_event) {  // the first nine products are point_layer[0..7] (build_dw_batch): reduce them, signal, then the rest
      HIP_TRY(launch_dw(batch, Mtot, slabs, st, 0, DW_EARLY_ITEMS));
      HIP_TRY(launch_dw_reduce(batch, st, 0, DW_EARLY_ITEMS));
      HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(early_event), st));
      HIP_TRY(launch_dw(batch, Mtot, slabs, st, DW_EARLY_ITEMS, batch.n - DW_EARLY_ITEMS));
      HIP_TRY(launch_dw_reduce(batch, st, DW_EARLY_ITEMS, batch.n - DW_EARLY_ITEMS));
    } else {
      HIP_TRY(launch_dw(batch, Mtot, slabs, st));
      HIP_TRY(launch_dw_reduce(batch, st));
    }
    FoldGradArgs fg;
    fg.M = at<float>(ws, L.mbuf); fg.db_dir = dw[B_DIR]; fg.w_dir = w.p[W_DIR]; fg.w_pi = w.p[W_PI]; fg.b_pi = w.p[B_PI];
    fg.dW_pi = dw[W_PI]; fg.db_pi = dw[B_PI]; fg.dW_dir = dw[W_DIR];
    HIP_TRY(launch_fold_grads(fg, st));
    // direction-encoding columns of dir_info (per-ray sums)
    SmallGradArgs sg;
    memset(&sg, 0, sizeof(sg));
    sg.save = save; sg.G = G; sg.dz = at<float>(ws, L.dz); sg.dspre = at<float>(ws, L.dspre); sg.rayf = at<float>(ws, L.rayf);
    sg.Mtot = Mtot; sg.MSrows = Mtot + DUMP_ROWS; sg.B = B; sg.Nc = Nc; sg.Nf = Nf;
    sg.dW_color = dw[W_COLOR]; sg.db_color = dw[B_COLOR]; sg.dw_sigma = dw[W_SIGMA]; sg.db_sigma = dw[B_SIGMA]; sg.dW_dir = dw[W_DIR];
    sg.sbuf = at<float>(ws, L.sbuf); sg.gdbuf = at<float>(ws, L.gdbuf); sg.sums_done = ray_duty ? 1 : 0;
    HIP_TRY(launch_small_grads(sg, slabs, st));  // (the slabs are free after the reduce: scratch of the gamma_d columns' two-step sum)
  }
  return NERF_HIP_OK;
}

}  // namespace

extern "C" {

int nerf_hip_train_step(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                        const float* K_inv9, const float* ray0_near_far, const float* C_true, int B, int Nc, int Nf, float last_delta,
                        float* C_coarse, float* C_fine, float* loss, float* const* dweights24, void* ws, size_t ws_bytes, int flags,
                        void* stream, void* early_event) {
  if (!C_true || !loss || !C_coarse || !C_fine) return fail(NERF_HIP_ERR_ARG, "null argument");
  flags |= NERF_HIP_SAVE_FOR_BACKWARD;
  if (int rc = check_weights(const_cast<const float* const*>(dweights24))) return rc;  // (before anything is enqueued)
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  if (!ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  const WsLayout L = layout(B, Nc, Nf, flags);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  TrainLoss tl;
  tl.C_true = C_true; tl.dC_c = at<float>(ws, L.dC); tl.dC_f = tl.dC_c + (size_t)B * 3; tl.terms = tl.dC_f + (size_t)B * 3; tl.loss = loss;
  tl.fused = false;
  if (int rc = forward_impl(weights24, row, col, poses_bound, K_inv9, ray0_near_far, B, Nc, Nf, last_delta, C_coarse, C_fine, ws, ws_bytes, flags,
                            stream, &tl))
    return rc;
  if (!tl.fused)  // (fused: d loss / d C and the loss's summands came out of the fine pass's epilogue; its sum is taken in the backward)
    if (int rc = nerf_hip_ray_loss(C_coarse, C_fine, C_true, B, loss, tl.dC_c, tl.dC_f, stream)) return rc;
  return backward_impl(weights24, tl.dC_c, tl.dC_f, ray0_near_far, B, Nc, Nf, last_delta, dweights24, ws, ws_bytes,
                       flags & ~NERF_HIP_WEIGHTS_UNCHANGED, stream, early_event, &tl);
}

int nerf_hip_ws_offset(int B, int Nc, int Nf, int flags, const char* name, size_t* offset) {
  if (!name || !offset) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  const WsLayout L = layout(B, Nc, Nf, flags);
  struct { const char* n; size_t o; } tab[] = {
      {"status", L.status}, {"dbg", L.dbg}, {"packed", L.packed}, {"packed_bf", L.packed_bf}, {"bsave", L.bsave}, {"bsave2", L.bsave2}, {"bmask", L.bmask}, {"bG", L.bG}, {"bG2", L.bG2}, {"rayf", L.rayf}, {"dvec", L.dvec}, {"t_c", L.t_c}, {"sig_c", L.sig_c},
      {"rgb_c", L.rgb_c}, {"w_c", L.w_c}, {"t_f", L.t_f}, {"sig_f", L.sig_f}, {"rgb_f", L.rgb_f}, {"perm", L.perm},
      {"w_m", L.w_m}, {"bundle", L.bundle}, {"save", L.save}, {"masks", L.masks}, {"spre", L.spre}, {"G", L.G}, {"dz", L.dz},
      {"dspre", L.dspre}, {"mbuf", L.mbuf}, {"fold", L.fold}, {"drgb_c", L.drgb_c}, {"dsig_c", L.dsig_c}, {"drgb_f", L.drgb_f}, {"dsig_f", L.dsig_f},
      {"dt_f", L.dt_f}, {"slabs", L.slabs}, {"masks", L.masks}, {"sbuf", L.sbuf}, {"gdbuf", L.gdbuf}};
  for (auto& e : tab)
    if (strcmp(e.n, name) == 0) {
      if (e.o == 0 && strcmp(name, "status") != 0) return fail(NERF_HIP_ERR_ARG, "buffer %s is not part of this layout (flags=%d)", name, flags);
      *offset = e.o;
      return NERF_HIP_OK;
    }
  return fail(NERF_HIP_ERR_ARG, "unknown workspace buffer %s", name);
}

int nerf_hip_ray_loss(const float* C_coarse, const float* C_fine, const float* C_true, int B, float* loss, float* dC_coarse,
                      float* dC_fine, void* stream) {
  if (!C_coarse || !C_fine || !C_true || !loss || B < 1) return fail(NERF_HIP_ERR_ARG, "null argument");
  HIP_TRY(launch_ray_loss(C_coarse, C_fine, C_true, B, loss, dC_coarse, dC_fine, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

int nerf_hip_read_status(const void* ws, size_t ws_bytes, uint32_t* status, void* stream) {
  if (!ws || !status || ws_bytes < 256) return fail(NERF_HIP_ERR_ARG, "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t w[STATUS_STICKY_WORD + 3];
  memset(w, 0, sizeof(w));
  HIP_TRY(hipMemcpyAsync(w, ws, sizeof(w), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // legacy scheme: word 0 (zeroed by the call's first kernel); stamped scheme (common.h STATUS_*): the flags count only with this call's stamp
  if (w[STATUS_SCHEME_WORD] == 1u) *status = ((w[STATUS_STAMPED_WORD] >> 8) == w[STATUS_GEN_WORD]) ? (w[STATUS_STAMPED_WORD] & 0xffu) : 0u;
  else *status = w[0] & ~(uint32_t)NERF_HIP_STATUS_PREP_TIMEOUT;
  // the one-launch preparation of the bf16-MLP calls (prep_bf16.hip): the weight image now in this workspace was packed by the call whose
  // token is word 34; word 33 = the last call whose wait for the fold timed out.  Equal and non-zero: the image is POISONED (NaN), and so is
  // every call that used or reuses it (NERF_HIP_WEIGHTS_UNCHANGED) until the next packing call
  if (w[STATUS_STICKY_WORD + 1] != 0u && w[STATUS_STICKY_WORD + 1] == w[STATUS_STICKY_WORD + 2]) *status |= NERF_HIP_STATUS_PREP_TIMEOUT;
  return NERF_HIP_OK;
}

int nerf_hip_read_status_sticky(void* ws, size_t ws_bytes, uint32_t* status, int clear, void* stream) {
  if (!ws || !status || ws_bytes < 256) return fail(NERF_HIP_ERR_ARG, "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t* word = static_cast<uint32_t*>(ws) + STATUS_STICKY_WORD;
  HIP_TRY(hipMemcpyAsync(status, word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  if (clear) HIP_TRY(hipMemsetAsync(word, 0, sizeof(uint32_t), st));
  HIP_TRY(hipStreamSynchronize(st));
  return NERF_HIP_OK;
}

int nerf_hip_rays(const int64_t* row, const int64_t* col, const float* poses_bound, const float* K_inv9, int B, int Nc,
                  float* d_cam, float* d_wrd, float* t_coarse, void* stream) {
  if (!row || !col || !poses_bound || !K_inv9 || B < 1 || Nc < 2) return fail(NERF_HIP_ERR_ARG, "bad argument");
  if (int rc = check_device()) return rc;
  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nc;
  ra.d_cam = d_cam; ra.d_wrd = d_wrd; ra.t_c = t_coarse;
  HIP_TRY(launch_rays(ra, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

int nerf_hip_field(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                   const float* K_inv9, const float* t, int B, int N, float* rgb, float* sigma, float* pts, float* gamma_p,
                   void* ws, size_t ws_bytes, void* stream) {
  if (B < 1 || N < 1 || N > 1024) return fail(NERF_HIP_ERR_ARG, "bad sizes");
  if (int rc = check_weights(weights24)) return rc;
  if (!row || !col || !poses_bound || !K_inv9 || !t || !rgb || !sigma || !ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  const int Nl = N < 2 ? 2 : N;
  const WsLayout L = layout(B < 2 ? 2 : B, Nl, Nl, 0);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  if (int rc = check_device()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Weights24 w = as_w24(weights24);
  HIP_TRY(launch_pack_weights(w, at<float>(ws, L.fold), at<float4>(ws, L.packed), NSEG_FWD, st));
  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nl;
  ra.rayf = at<float>(ws, L.rayf);
  ra.dvec = at<float>(ws, L.dvec);
  ra.w_dir = w.p[W_DIR]; ra.b_dir = w.p[B_DIR]; ra.b_fold = at<float>(ws, L.fold);
  HIP_TRY(launch_rays(ra, st));
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.wp = at<float4>(ws, L.packed);
  fa.w = w;
  fa.rayf = at<float>(ws, L.rayf);
  fa.dvec = at<float>(ws, L.dvec);
  fa.t = t; fa.rgb = rgb; fa.sigma = sigma; fa.pts_dbg = pts; fa.gp_dbg = gamma_p;
  fa.N = N; fa.M = B * N;
  HIP_TRY(launch_field_fwd_reg(fa, false, st));
  return NERF_HIP_OK;
}

int nerf_hip_field_bf16(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                        const float* K_inv9, const float* t, int B, int N, float* rgb, float* sigma, void* ws, size_t ws_bytes,
                        void* stream) {
  if (B < 1 || N < 1 || N > 1024) return fail(NERF_HIP_ERR_ARG, "bad sizes");
  if (int rc = check_weights(weights24)) return rc;
  if (!row || !col || !poses_bound || !K_inv9 || !t || !rgb || !sigma || !ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  const int Nl = N < 2 ? 2 : N;
  const WsLayout L = layout(B < 2 ? 2 : B, Nl, Nl, NERF_HIP_BF16_MLP);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  if (int rc = check_device()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Weights24 w = as_w24(weights24);
  HIP_TRY(launch_fold_weights(w, at<float>(ws, L.fold), st));
  HIP_TRY(launch_pack_weights_bf16(w, at<float>(ws, L.fold), at<unsigned char>(ws, L.packed_bf), st));
  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nl;
  ra.rayf = at<float>(ws, L.rayf);
  HIP_TRY(launch_rays(ra, st));
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.wbf = at<unsigned char>(ws, L.packed_bf);
  fa.w = w;
  fa.rayf = at<float>(ws, L.rayf);
  fa.t = t; fa.rgb = rgb; fa.sigma = sigma;
  fa.N = N; fa.M = B * N;
  HIP_TRY(launch_field_fwd_bf16(fa, false, st));
  return NERF_HIP_OK;
}

int nerf_hip_coarse_composite(const float* t_c, const float* sigma_c, const float* rgb_c, const float* near_far, float delta0,
                              int B, int Nc, int Nf, float* w_c, float* C_coarse, float* t_f, uint32_t* status, void* stream) {
  if (!t_c || !sigma_c || !rgb_c || !near_far || !t_f) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (B < 1 || Nc < 2 || Nc > 1024 || Nf < 1 || Nf > 1024) return fail(NERF_HIP_ERR_ARG, "bad sizes");
  CoarseArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.t_c = t_c; ca.sigma = sigma_c; ca.rgb = rgb_c; ca.near_far = near_far;
  ca.B = B; ca.Nc = Nc; ca.Nf = Nf;
  ca.delta0_mode = 1; ca.delta0 = delta0;
  ca.w_c = w_c; ca.C_coarse = C_coarse; ca.t_f = t_f; ca.status = status;
  HIP_TRY(launch_coarse(ca, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

static const int kParamNumel[24] = {256 * 60, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 316, 256, 256 * 256, 256,
                                     256 * 256, 256, 256 * 256, 256, 256, 1, 256 * 256, 256, 128 * 280, 128, 3 * 128, 3};

int nerf_hip_adam_step(float* const* params24, const float* const* grads24, float* exp_avg, float* exp_avg_sq, int step, float lr,
                       float beta1, float beta2, float eps, void* stream) {
  if (!params24 || !grads24 || !exp_avg || !exp_avg_sq || step < 1) return fail(NERF_HIP_ERR_ARG, "bad argument");
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  int off = 0;
  for (int i = 0; i < 24; ++i) {
    if (!params24[i] || !grads24[i]) return fail(NERF_HIP_ERR_ARG, "null tensor %d", i);
    a.param[i] = params24[i]; a.grad[i] = grads24[i]; a.numel[i] = kParamNumel[i]; a.offset[i] = off;
    off += kParamNumel[i];
  }
  a.m = exp_avg; a.v = exp_avg_sq;
  // bias corrections in double like torch's Python scalars
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.bias2_sqrt = (float)sqrt(bc2);
  a.one_minus_beta1 = 1.0f - beta1; a.beta2 = beta2; a.one_minus_beta2 = 1.0f - beta2; a.eps = eps;
  HIP_TRY(launch_adam(a, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

int nerf_hip_gather_rays(const int64_t* index, const float* pixels, const float* poses17, int B, int H, int W, int64_t* row,
                         int64_t* col, int64_t* pic, float* pix_val, float* poses_bound, void* stream) {
  if (!index || !pixels || !poses17 || !row || !col || !pic || !pix_val || !poses_bound || B < 1 || H < 1 || W < 1)
    return fail(NERF_HIP_ERR_ARG, "bad argument");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.index = reinterpret_cast<const long long*>(index); g.pixels = pixels; g.poses = poses17; g.B = B; g.H = H; g.W = W;
  g.row = reinterpret_cast<long long*>(row); g.col = reinterpret_cast<long long*>(col); g.pic = reinterpret_cast<long long*>(pic);
  g.pix_val = pix_val; g.poses_bound = poses_bound;
  HIP_TRY(launch_gather_rays(g, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

int nerf_hip_coarse_composite_backward(const float* t_c, const float* sigma_c, const float* rgb_c, const float* near_far,
                                       float delta0, int B, int Nc, int Nf, const float* dC_coarse, const float* dt_f,
                                       float* dsig_c, float* drgb_c, void* stream) {
  if (!t_c || !sigma_c || !rgb_c || !near_far || !dC_coarse || !dt_f || !dsig_c || !drgb_c) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (B < 1 || Nc < 2 || Nc > 1024 || Nf < 1 || Nf > 1024) return fail(NERF_HIP_ERR_ARG, "bad sizes");
  CoarseBwdArgs cb;
  memset(&cb, 0, sizeof(cb));
  cb.dC_c = dC_coarse; cb.dt_f = dt_f; cb.t_c = t_c; cb.sigma = sigma_c; cb.rgb = rgb_c; cb.near_far = near_far;
  cb.B = B; cb.Nc = Nc; cb.Nf = Nf; cb.delta0_mode = 1; cb.delta0 = delta0;
  cb.drgb_c = drgb_c; cb.dsig_c = dsig_c;
  HIP_TRY(launch_coarse_bwd(cb, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

int nerf_hip_merge_composite(const float* t_c, const float* t_f, const float* sigma_c, const float* sigma_f, const float* rgb_c,
                             const float* rgb_f, int B, int Nc, int Nf, float last_delta, float* bundle, float* w, float* C_fine,
                             void* stream) {
  if (!t_c || !t_f || !sigma_c || !sigma_f || !rgb_c || !rgb_f || !C_fine) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (B < 1 || Nc < 1 || Nf < 1 || Nc + Nf > 2048) return fail(NERF_HIP_ERR_ARG, "bad sizes");
  MergeArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.t_c = t_c; ma.t_f = t_f; ma.sig_c = sigma_c; ma.sig_f = sigma_f; ma.rgb_c = rgb_c; ma.rgb_f = rgb_f;
  ma.B = B; ma.Nc = Nc; ma.Nf = Nf; ma.P = next_pow2(Nc + Nf); ma.last = last_delta;
  ma.bundle = bundle; ma.w = w; ma.C_fine = C_fine;
  HIP_TRY(launch_merge(ma, static_cast<hipStream_t>(stream)));
  return NERF_HIP_OK;
}

}  // extern "C"
