// api.hip -- the extern "C" surface of libnerf_hip.so (include/nerf_hip.h).  Host code only: argument
// checks, workspace carve-up and kernel sequencing on the caller's stream.  No allocation, no host sync
// (except nerf_hip_read_status).
#include "../../include/nerf_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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
  return NERF_HIP_OK;
}

WsLayout layout(int B, int Nc, int Nf, int flags) {
  WsLayout L;
  memset(&L, 0, sizeof(L));
  const size_t b = (size_t)B, N = (size_t)Nc + Nf;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
  L.status = take(256);
  L.packed = take((size_t)PACKED_ALL_F4 * 16);
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
    L.save_c = take((size_t)10 * b * Nc * WIDTH * 4);
    L.save_f = take((size_t)10 * b * Nf * WIDTH * 4);
    L.spre_c = take(b * Nc * 4);
    L.spre_f = take(b * Nf * 4);
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
struct Prof {
  bool on = false;
  int cap = 0, used = 0;
  hipEvent_t* ev = nullptr;  // 2 per launch
  int* kid = nullptr;
} g_prof;

struct ProfScope {
  hipStream_t st;
  int slot = -1;
  ProfScope(int kernel_id, hipStream_t s) : st(s) {
    if (g_prof.on && g_prof.used < g_prof.cap) {
      slot = g_prof.used++;
      g_prof.kid[slot] = kernel_id;
      (void)hipEventRecord(g_prof.ev[2 * slot], st);
    }
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
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

int nerf_hip_forward(const float* const* weights24, const int64_t* row, const int64_t* col, const float* poses_bound,
                     const float* K_inv9, const float* ray0_near_far, int B, int Nc, int Nf, float last_delta, float* C_coarse,
                     float* C_fine, void* ws, size_t ws_bytes, int flags, void* stream) {
  if (int rc = check_sizes(B, Nc, Nf)) return rc;
  if (int rc = check_weights(weights24)) return rc;
  if (!row || !col || !poses_bound || !K_inv9 || !C_coarse || !C_fine || !ws) return fail(NERF_HIP_ERR_ARG, "null argument");
  if (((uintptr_t)ws & 255) != 0) return fail(NERF_HIP_ERR_ARG, "workspace must be 256-byte aligned");
  const WsLayout L = layout(B, Nc, Nf, flags);
  if (ws_bytes < L.total) return fail(NERF_HIP_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  if (int rc = check_device()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool save = (flags & NERF_HIP_SAVE_FOR_BACKWARD) != 0;
  const Weights24 w = as_w24(weights24);

  HIP_TRY(hipMemsetAsync(at<void>(ws, L.status), 0, 256, st));
  { ProfScope ps(NERF_HIP_K_PACK, st); HIP_TRY(launch_pack_weights(w, at<float4>(ws, L.packed), save ? NSEG : NSEG_FWD, st)); }

  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nc;
  ra.rayf = at<float>(ws, L.rayf);
  ra.dvec = at<float>(ws, L.dvec);
  ra.w_dir = w.p[W_DIR]; ra.b_dir = w.p[B_DIR];
  ra.t_c = at<float>(ws, L.t_c);
  { ProfScope ps(NERF_HIP_K_RAYS, st); HIP_TRY(launch_rays(ra, st)); }

  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.wp = at<float4>(ws, L.packed);
  fa.w = w;
  fa.rayf = at<float>(ws, L.rayf);
  fa.dvec = at<float>(ws, L.dvec);
  // coarse pass (nerf.py:289)
  fa.t = at<float>(ws, L.t_c);
  fa.rgb = at<float>(ws, L.rgb_c);
  fa.sigma = at<float>(ws, L.sig_c);
  fa.N = Nc; fa.M = B * Nc;
  if (save) { fa.save = at<float>(ws, L.save_c); fa.spre = at<float>(ws, L.spre_c); }
  { ProfScope ps(NERF_HIP_K_FIELD_COARSE, st); HIP_TRY(launch_field_fwd(fa, save, st)); }

  CoarseArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.t_c = at<float>(ws, L.t_c); ca.sigma = at<float>(ws, L.sig_c); ca.rgb = at<float>(ws, L.rgb_c);
  ca.rayf = at<float>(ws, L.rayf);
  ca.B = B; ca.Nc = Nc; ca.Nf = Nf;
  ca.delta0_mode = 0;
  if (ray0_near_far) { ca.ray0_override = 1; ca.near0 = ray0_near_far[0]; ca.far0 = ray0_near_far[1]; }
  ca.w_c = at<float>(ws, L.w_c); ca.C_coarse = C_coarse; ca.t_f = at<float>(ws, L.t_f);
  ca.status = at<uint32_t>(ws, L.status);
  { ProfScope ps(NERF_HIP_K_COARSE, st); HIP_TRY(launch_coarse(ca, st)); }

  // fine pass (nerf.py:299), same network (quirk Q10)
  fa.t = at<float>(ws, L.t_f);
  fa.rgb = at<float>(ws, L.rgb_f);
  fa.sigma = at<float>(ws, L.sig_f);
  fa.N = Nf; fa.M = B * Nf;
  if (save) { fa.save = at<float>(ws, L.save_f); fa.spre = at<float>(ws, L.spre_f); }
  { ProfScope ps(NERF_HIP_K_FIELD_FINE, st); HIP_TRY(launch_field_fwd(fa, save, st)); }

  MergeArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.t_c = at<float>(ws, L.t_c); ma.t_f = at<float>(ws, L.t_f);
  ma.sig_c = at<float>(ws, L.sig_c); ma.sig_f = at<float>(ws, L.sig_f);
  ma.rgb_c = at<float>(ws, L.rgb_c); ma.rgb_f = at<float>(ws, L.rgb_f);
  ma.B = B; ma.Nc = Nc; ma.Nf = Nf; ma.P = next_pow2(Nc + Nf);
  ma.last = last_delta;
  if (save) { ma.bundle = at<float>(ws, L.bundle); ma.w = at<float>(ws, L.w_m); ma.perm = at<uint16_t>(ws, L.perm); }
  ma.C_fine = C_fine;
  { ProfScope ps(NERF_HIP_K_MERGE, st); HIP_TRY(launch_merge(ma, st)); }
  return NERF_HIP_OK;
}

int nerf_hip_profile_begin(int max_launches) {
  if (g_prof.ev) return fail(NERF_HIP_ERR_ARG, "profile already active");
  if (max_launches < 1 || max_launches > (1 << 20)) return fail(NERF_HIP_ERR_ARG, "bad max_launches");
  g_prof.ev = new hipEvent_t[2 * (size_t)max_launches];
  g_prof.kid = new int[max_launches];
  for (int i = 0; i < 2 * max_launches; ++i) HIP_TRY(hipEventCreate(&g_prof.ev[i]));
  g_prof.cap = max_launches;
  g_prof.used = 0;
  g_prof.on = true;
  return NERF_HIP_OK;
}

int nerf_hip_profile_end(double* ms_sum, int* count, int n_kernels) {
  if (!g_prof.ev) return fail(NERF_HIP_ERR_ARG, "profile not active");
  g_prof.on = false;
  for (int k = 0; k < n_kernels; ++k) { if (ms_sum) ms_sum[k] = 0.0; if (count) count[k] = 0; }
  int rc = NERF_HIP_OK;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    if (e != hipSuccess) { rc = fail(NERF_HIP_ERR_DEVICE, "profile event: %s", hipGetErrorString(e)); break; }
    const int k = g_prof.kid[i];
    if (k >= 0 && k < n_kernels) { if (ms_sum) ms_sum[k] += ms; if (count) count[k] += 1; }
  }
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  delete[] g_prof.kid;
  g_prof.ev = nullptr; g_prof.kid = nullptr; g_prof.cap = g_prof.used = 0;
  return rc;
}

int nerf_hip_backward(const float* const* weights24, const float* dC_coarse, const float* dC_fine, int B, int Nc, int Nf,
                      float last_delta, float* const* dweights24, void* ws, size_t ws_bytes, int flags, void* stream) {
  (void)weights24; (void)dC_coarse; (void)dC_fine; (void)B; (void)Nc; (void)Nf; (void)last_delta; (void)dweights24;
  (void)ws; (void)ws_bytes; (void)flags; (void)stream;
  return fail(NERF_HIP_ERR_ARG, "nerf_hip_backward: not built yet");
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
  HIP_TRY(hipMemcpyAsync(status, ws, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
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
  HIP_TRY(launch_pack_weights(w, at<float4>(ws, L.packed), NSEG_FWD, st));
  RaysArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.row = row; ra.col = col; ra.pb = poses_bound;
  memcpy(ra.K, K_inv9, 9 * sizeof(float));
  ra.B = B; ra.Nc = Nl;
  ra.rayf = at<float>(ws, L.rayf);
  ra.dvec = at<float>(ws, L.dvec);
  ra.w_dir = w.p[W_DIR]; ra.b_dir = w.p[B_DIR];
  HIP_TRY(launch_rays(ra, st));
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.wp = at<float4>(ws, L.packed);
  fa.w = w;
  fa.rayf = at<float>(ws, L.rayf);
  fa.dvec = at<float>(ws, L.dvec);
  fa.t = t; fa.rgb = rgb; fa.sigma = sigma; fa.pts_dbg = pts; fa.gp_dbg = gamma_p;
  fa.N = N; fa.M = B * N;
  HIP_TRY(launch_field_fwd(fa, false, st));
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
