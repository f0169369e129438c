// field_fwd.hip -- weight packing + the LDS-tile form of the fused field query for MI355X (gfx950) (A/B reference,
// NERF_HIP_FORCE_TILE_KERNEL; the product path is the register-resident form in field_fwd_reg.hip): sample point -> positional encoding ->
// 8x256 MLP (+ sigma / feature / direction / colour heads) for a tile of 64 samples per workgroup.
//
// Replaces, per sample: nerf.py:200-216 (world point), Encoder.forward nerf.py:135-167,
// Network.forward nerf.py:101-124.
//
// Design:
//  * one workgroup = 256 threads = 4 waves owns 64 samples; their activations [64][256] f32 live in
//    LDS for the whole network (row stride 260 floats: conflict-free ds_read_b128 / ds_write_b128);
//  * every 256-wide layer is computed as  D[feature][sample] = W[feature][k] * act[sample][k]  with
//    v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain): the WEIGHTS are the A operand, the
//    activations the B operand, so a lane's 16 accumulators are 4x4 consecutive features of ONE
//    sample and go back to LDS as four 16-byte stores, already in the layout the next layer reads;
//  * wave w owns features [64w, 64w+64) x 64 samples = 2x2 MFMA tiles (64 accumulator VGPRs);
//  * weights are pre-packed in fragment order (k_pack_weights) so a lane fetches 4 k-steps of its
//    A fragment with ONE coalesced 16-byte global load (1 KiB per wave instruction, L2 resident:
//    all 2.36 MB of weights fit every XCD's 4 MiB L2); the next k-block's fragment is prefetched
//    behind the current block's 16 MFMAs (1024 cycles), which hides the L2 latency;
//  * two workgroups per CU (2 x 68 KiB LDS): one's barriers/epilogues hide under the other's MFMAs.
#include "field_common.h"
#include "prep_parts.h"

namespace nerf {




// ------------------------------------------------------------------------------------------
// weight packing: raw nn.Linear [out][in] -> MFMA A-fragment order, forward and transposed.
// ------------------------------------------------------------------------------------------
struct PackDesc { int src; int ld; int col0; int rows; int cols; int transposed; };
// element (feature f, input k) of segment s:
//   forward:     W[src][f][col0 + k]           f < rows, k < cols
//   transposed:  W[src][k][col0 + f]           f < cols(of slice) ...
__device__ __forceinline__ PackDesc pack_desc(int s) {
  switch (s) {
    case SEG_L0: return {0, 60, 0, 256, 60, 0};
    case SEG_L1: return {2, 256, 0, 256, 256, 0};
    case SEG_L2: return {4, 256, 0, 256, 256, 0};
    case SEG_L3: return {6, 256, 0, 256, 256, 0};
    case SEG_L4A: return {8, 316, 0, 256, 256, 0};
    case SEG_L4B: return {8, 316, 256, 256, 60, 0};
    case SEG_L5: return {10, 256, 0, 256, 256, 0};
    case SEG_L6: return {12, 256, 0, 256, 256, 0};
    case SEG_L7: return {14, 256, 0, 256, 256, 0};
    case SEG_PI: return {W_PI, 256, 0, 256, 256, 0};
    case SEG_DIR: return {W_DIR, 280, 24, 128, 256, 0};
    case SEG_FOLD: return {24, 256, 0, 128, 256, 0};
    case SEG_T_FOLD: return {24, 256, 0, 256, 128, 1};
    // transposed: rows = number of output features of the transposed product (= inputs of the layer),
    // cols = reduction length (= outputs of the layer)
    case SEG_T_DIR: return {W_DIR, 280, 24, 256, 128, 1};
    case SEG_T_PI: return {W_PI, 256, 0, 256, 256, 1};
    case SEG_T_L7: return {14, 256, 0, 256, 256, 1};
    case SEG_T_L6: return {12, 256, 0, 256, 256, 1};
    case SEG_T_L5: return {10, 256, 0, 256, 256, 1};
    case SEG_T_L4A: return {8, 316, 0, 256, 256, 1};
    case SEG_T_L4B: return {8, 316, 256, 60, 256, 1};
    case SEG_T_L3: return {6, 256, 0, 256, 256, 1};
    case SEG_T_L2: return {4, 256, 0, 256, 256, 1};
    case SEG_T_L1: return {2, 256, 0, 256, 256, 1};
    default: return {0, 60, 0, 60, 256, 1};  // SEG_T_L0
  }
}

// first float4 of every segment (seg_off4 evaluated at compile time: called per thread it is a loop inside a loop)
struct SegOffTable { int v[NSEG + 1]; };
constexpr SegOffTable make_seg_off_table() {
  SegOffTable t{};
  for (int s = 0; s <= NSEG; ++s) t.v[s] = seg_off4(s);
  return t;
}
__constant__ const SegOffTable kSegOff = make_seg_off_table();

// The fold, in front of every packer (fp32 and bf16-MLP variant alike): fold[0 .. 128) = b_fold = W_dir[:, 24:] b_pi (k_rays adds it to
// every ray's dir_info start vector), fold[128 + o * 256 + k] = W_fold[o][k] = sum_j W_dir[o][24 + j] * W_pi[j][k] (common.h SEG_FOLD).
// 129 blocks: block o < 128 = row o of W_fold, thread (part, k4): four fp32 fma chains over a quarter of the j range each (the W_dir
// element of a step is wave-uniform, the W_pi row a coalesced KiB), the quarters added in a fixed order; block 128 = b_fold.
__global__ __launch_bounds__(256) void k_fold_weights(Weights24 w, float* __restrict__ fold) { fold_block(w, fold, blockIdx.x); }

hipError_t launch_fold_weights(const Weights24& w, float* fold, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_weights, dim3(HALF + 1), dim3(256), 0, st, w, fold);
  return hipGetLastError();
}

// The folded segments (SEG_FOLD, SEG_T_FOLD) are packed from the fp32 W_fold that k_fold_weights left in the workspace (launched in
// front of this kernel on the same stream): forward, transposed and -- in the bf16 variant -- rounded images all come from ONE matrix.
__global__ __launch_bounds__(256) void k_pack_weights(Weights24 w, float4* __restrict__ out, int nseg, const float* __restrict__ fold) {
  // one thread per float4 of the packed image
  const int total = kSegOff.v[nseg];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int s = 0;
#pragma unroll 1
    while (s + 1 < nseg && idx >= kSegOff.v[s + 1]) ++s;
    const int local = idx - kSegOff.v[s];
    const int kbn = seg_kb(s);
    const int lane = local & 63;
    const int kb = (local >> 6) % kbn;
    const int ft = (local >> 6) / kbn;
    const int f = ft * 32 + (lane & 31);
    const int k0 = kb * 8 + 4 * (lane >> 5);
    const PackDesc d = pack_desc(s);
    const float* W = d.src < 24 ? w.p[d.src] : fold + HALF;  // src 24: W_fold[128][256]
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!d.transposed) {
      // four consecutive inputs of one output row: one 16-byte load (every row length and column offset is a multiple of 4 floats,
      // and so is every segment's input count, so a group is either all inside or all padding)
      if (f < d.rows && k0 < d.cols) {
        const float* src = W + (size_t)f * d.ld + d.col0 + k0;
        if ((reinterpret_cast<uintptr_t>(W) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
        else v = make_float4(src[0], src[1], src[2], src[3]);  // a caller's parameter view that is only 4-byte aligned
      }
    } else if (f < d.rows) {
      if (k0 + 0 < d.cols) v.x = W[(size_t)(k0 + 0) * d.ld + d.col0 + f];
      if (k0 + 1 < d.cols) v.y = W[(size_t)(k0 + 1) * d.ld + d.col0 + f];
      if (k0 + 2 < d.cols) v.z = W[(size_t)(k0 + 2) * d.ld + d.col0 + f];
      if (k0 + 3 < d.cols) v.w = W[(size_t)(k0 + 3) * d.ld + d.col0 + f];
    }
    out[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------
// the fused forward kernel
// ------------------------------------------------------------------------------------------
#ifdef NERF_STAMPS
#define STAMP(slot)                                                   \
  do {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                \
    const unsigned long long t_ = __builtin_readcyclecounter();       \
    __builtin_amdgcn_sched_barrier(0);                                \
    tsum[slot] += t_ - tlast;                                         \
    tlast = t_;                                                       \
  } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void k_field_fwd(const FieldArgs a) {
#ifdef NERF_STAMPS
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;
  float* scr = smem + TM * LDA;  // [4][64][3]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * TM;
  const int sm = lane;  // sample handled by this thread in the per-sample (VALU) stages
  const int m = m0 + sm;
  const bool valid = m < a.M;
  const int mc = valid ? m : a.M - 1;
  const int ray = mc / a.N;
  const float* rf = a.rayf + (size_t)ray * RAYF;
  const int nrows = (a.M - m0) < TM ? (a.M - m0) : TM;  // live rows of this tile
  const long long grow0 = (long long)a.row0 + m0;       // first row in the combined (coarse|fine) buffers
  const size_t MS = (size_t)a.MSrows * WIDTH;              // stride between saved tensors
  uint16_t* mk = SAVE ? a.masks + ((size_t)(a.tile0 + blockIdx.x) * 4) * 256 + tid : nullptr;
  const size_t MKS = (size_t)a.tiles_tot * 4 * 256;      // stride between mask layers

  __builtin_amdgcn_s_setprio(2);
  float p[3];
  sample_point(rf, a.t[mc], p);
  if (a.pts_dbg && valid && wv == 0) {
    a.pts_dbg[(size_t)m * 3 + 0] = p[0];
    a.pts_dbg[(size_t)m * 3 + 1] = p[1];
    a.pts_dbg[(size_t)m * 3 + 2] = p[2];
  }
  encode_point_to_lds(p, act, sm, wv);
  STAMP(0);
  __syncthreads();
  STAMP(4);
  if (a.gp_dbg) {
    for (int idx = tid; idx < TM * POINT_DIM; idx += 256) {
      const int r = idx / POINT_DIM, c = idx - r * POINT_DIM;
      if (m0 + r < a.M) a.gp_dbg[(size_t)(m0 + r) * POINT_DIM + c] = act[r * LDA + c];
    }
  }

  if (SAVE) save_rows(act, a.save + S_GP * MS, grow0, nrows, 16, WIDTH, tid);

  f32x16 acc[2][2];
  const int fbase = wv * 64;

  // Each layer's first weight fragments are requested before the previous layer's barriers and epilogue.
  WFrag<2> wc;
  float4 bq[2][4];
  // ---- layer 0: 60(64) -> 256
  acc_init_bias<2>(a.w.p[B_L0], fbase, lane, acc);
  __builtin_amdgcn_s_setprio(0);
  mfma_layer<8, 2>(a.wp + seg_off4(SEG_L0), wv * 2, act, 0, lane, acc);
  __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
  wfrag_first<32, 2>(a.wp + seg_off4(SEG_L1), wv * 2, lane, wc);
  bias_first<2>(a.w.p[3], fbase, lane, bq);
  __syncthreads();
  STAMP(2);
  acc_store<2, true>(act, fbase, lane, acc, mk);
  STAMP(3);
  __syncthreads();
  STAMP(4);
  if (SAVE) save_rows(act, a.save + 0 * MS, grow0, nrows, 64, WIDTH, tid);

  // ---- layers 1..3
#pragma unroll 1
  for (int l = 1; l <= 3; ++l) {
    acc_init_regs<2>(bq, acc);
    __builtin_amdgcn_s_setprio(0);
    mfma_layer<32, 2>(a.wp + seg_off4(SEG_L1) + (size_t)(l - 1) * 8 * 32 * 64, wv * 2, act, 0, lane, acc, wc);
    __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
    // next: layer l+1 (L2, L3) or L4A -- consecutive segments of equal size
    wfrag_first<32, 2>(a.wp + seg_off4(SEG_L1) + (size_t)l * 8 * 32 * 64, wv * 2, lane, wc);
    bias_first<2>(a.w.p[2 * (l + 1) + 1], fbase, lane, bq);
    __syncthreads();
  STAMP(2);
    acc_store<2, true>(act, fbase, lane, acc, SAVE ? mk + l * MKS : nullptr);
  STAMP(3);
    __syncthreads();
  STAMP(4);
    if (SAVE) save_rows(act, a.save + (size_t)l * MS, grow0, nrows, 64, WIDTH, tid);
  }

  // ---- layer 4: cat(h3, gamma_p) (hidden first, nerf.py:109) -> 256
  acc_init_regs<2>(bq, acc);
  __builtin_amdgcn_s_setprio(0);
  mfma_layer<32, 2>(a.wp + seg_off4(SEG_L4A), wv * 2, act, 0, lane, acc, wc);
  __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
  wfrag_first<8, 2>(a.wp + seg_off4(SEG_L4B), wv * 2, lane, wc);
  __syncthreads();
  STAMP(2);
  encode_point_to_lds(p, act, sm, wv);  // re-encode into the (now free) activation buffer
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  mfma_layer<8, 2>(a.wp + seg_off4(SEG_L4B), wv * 2, act, 0, lane, acc, wc);
  __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
  wfrag_first<32, 2>(a.wp + seg_off4(SEG_L5), wv * 2, lane, wc);
  bias_first<2>(a.w.p[11], fbase, lane, bq);
  __syncthreads();
  STAMP(2);
  acc_store<2, true>(act, fbase, lane, acc, SAVE ? mk + 4 * MKS : nullptr);
  STAMP(3);
  __syncthreads();
  STAMP(4);
  if (SAVE) save_rows(act, a.save + 4 * MS, grow0, nrows, 64, WIDTH, tid);

  // ---- layers 5..7
#pragma unroll 1
  for (int l = 5; l <= 7; ++l) {
    acc_init_regs<2>(bq, acc);
    __builtin_amdgcn_s_setprio(0);
    mfma_layer<32, 2>(a.wp + seg_off4(SEG_L5) + (size_t)(l - 5) * 8 * 32 * 64, wv * 2, act, 0, lane, acc, wc);
    __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
    // next: L6, L7 or point_info -- consecutive segments of equal size
    wfrag_first<32, 2>(a.wp + seg_off4(SEG_L5) + (size_t)(l - 4) * 8 * 32 * 64, wv * 2, lane, wc);
    if (l < 7) bias_first<2>(a.w.p[2 * (l + 1) + 1], fbase, lane, bq);
    __syncthreads();
  STAMP(2);
    acc_store<2, true>(act, fbase, lane, acc, SAVE ? mk + l * MKS : nullptr);
  STAMP(3);
    __syncthreads();
  STAMP(4);
    if (SAVE) save_rows(act, a.save + (size_t)l * MS, grow0, nrows, 64, WIDTH, tid);
  }

  // ---- sigma head (VALU): sigma = |w_sigma . h7 + b|   (nerf.py:94, 115)
  {
    const float* ws = a.w.p[W_SIGMA] + wv * 64;
    const float* hr = act + sm * LDA + wv * 64;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(hr + k);
      s = __builtin_fmaf(hv.x, ws[k], s);
      s = __builtin_fmaf(hv.y, ws[k + 1], s);
      s = __builtin_fmaf(hv.z, ws[k + 2], s);
      s = __builtin_fmaf(hv.w, ws[k + 3], s);
    }
    scr[wv * 64 + sm] = s;
  }

  // ---- point_info: 256 -> 256, no activation (nerf.py:96, 117).  Its bias enters through the per-ray start vector of dir_info
  // (dvec includes W_dir[:, 24:] b_pi, common.h SEG_FOLD), so the accumulators start at zero here
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][st][r] = 0.f;
  __builtin_amdgcn_s_setprio(0);
  mfma_layer<32, 2>(a.wp + seg_off4(SEG_PI), wv * 2, act, 0, lane, acc, wc);
  __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
  WFrag<1> wd;
  wfrag_first<32, 1>(a.wp + seg_off4(SEG_DIR), wv, lane, wd);
  // dir_info: the gamma_d part (+ bias) of the pre-activation is per ray (dvec); requested before the barrier as well
  float4 dq[2][4];
  {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      int ms = m0 + st * 32 + j;
      ms = ms < a.M ? ms : a.M - 1;
      const float* dv = a.dvec + (size_t)(ms / a.N) * HALF + wv * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) dq[st][g] = *reinterpret_cast<const float4*>(dv + 8 * g);
    }
  }
  __syncthreads();
  STAMP(2);
  if (wv == 0 && valid) {
    const float pre = ((scr[sm] + scr[64 + sm]) + (scr[128 + sm] + scr[192 + sm])) + a.w.p[B_SIGMA][0];
    a.sigma[m] = fabsf(pre);
    if (SAVE) a.spre[a.row0 + m] = pre;
  }
  acc_store<2, false>(act, fbase, lane, acc);
  STAMP(3);
  __syncthreads();
  STAMP(4);

  // ---- dir_info: cat(gamma_d, feat) -> 128, ReLU (nerf.py:98, 118)
  f32x16 acd[1][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acd[0][st][4 * g + 0] = dq[st][g].x;
      acd[0][st][4 * g + 1] = dq[st][g].y;
      acd[0][st][4 * g + 2] = dq[st][g].z;
      acd[0][st][4 * g + 3] = dq[st][g].w;
    }
  __builtin_amdgcn_s_setprio(0);
  mfma_layer<32, 1>(a.wp + seg_off4(SEG_DIR), wv, act, 0, lane, acd, wd);
  __builtin_amdgcn_s_setprio(2);  // VALU/LDS phases outrank the partner workgroup's MFMA stream
  STAMP(1);
  __syncthreads();
  STAMP(2);
  acc_store<1, true>(act, wv * 32, lane, acd);
  STAMP(3);
  __syncthreads();
  STAMP(4);
  if (SAVE) save_rows(act, a.save + S_C * MS, grow0, nrows, 32, WIDTH, tid);

  // ---- colour head (VALU): rgb = sigmoid(W_c c + b)   (nerf.py:99, 119)
  {
    const float* wc = a.w.p[W_COLOR] + wv * 32;
    const float* cr = act + sm * LDA + wv * 32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 cv = *reinterpret_cast<const float4*>(cr + k);
      s0 = __builtin_fmaf(cv.x, wc[k], s0);
      s1 = __builtin_fmaf(cv.x, wc[HALF + k], s1);
      s2 = __builtin_fmaf(cv.x, wc[2 * HALF + k], s2);
      s0 = __builtin_fmaf(cv.y, wc[k + 1], s0);
      s1 = __builtin_fmaf(cv.y, wc[HALF + k + 1], s1);
      s2 = __builtin_fmaf(cv.y, wc[2 * HALF + k + 1], s2);
      s0 = __builtin_fmaf(cv.z, wc[k + 2], s0);
      s1 = __builtin_fmaf(cv.z, wc[HALF + k + 2], s1);
      s2 = __builtin_fmaf(cv.z, wc[2 * HALF + k + 2], s2);
      s0 = __builtin_fmaf(cv.w, wc[k + 3], s0);
      s1 = __builtin_fmaf(cv.w, wc[HALF + k + 3], s1);
      s2 = __builtin_fmaf(cv.w, wc[2 * HALF + k + 3], s2);
    }
    scr[(wv * 64 + sm) * 3 + 0] = s0;
    scr[(wv * 64 + sm) * 3 + 1] = s1;
    scr[(wv * 64 + sm) * 3 + 2] = s2;
  }
  __syncthreads();
  if (tid < 192) {
    const int s = tid / 3, ch = tid - 3 * s;
    if (m0 + s < a.M) {
      const float z = ((scr[s * 3 + ch] + scr[(64 + s) * 3 + ch]) + (scr[(128 + s) * 3 + ch] + scr[(192 + s) * 3 + ch])) + a.w.p[B_COLOR][ch];
      a.rgb[(size_t)(m0 + s) * 3 + ch] = 1.0f / (1.0f + expf(-z));
    }
  }
#ifdef NERF_STAMPS
  STAMP(5);
  if (a.stamps && lane == 0 && wv == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(a.stamps + i, tsum[i]);
    atomicAdd(a.stamps + 7, 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------
// launchers (called from api.cpp)
// ------------------------------------------------------------------------------------------
hipError_t launch_pack_weights(const Weights24& w, float* fold, float4* out, int nseg, hipStream_t st) {
  if (hipError_t e = launch_fold_weights(w, fold, st)) return e;
  const int total = seg_off4(nseg);
  hipLaunchKernelGGL(k_pack_weights, dim3((total + 255) / 256), dim3(256), 0, st, w, out, nseg, fold);
  return hipGetLastError();
}

hipError_t launch_field_fwd(const FieldArgs& a_in, bool save, hipStream_t st) {
  const FieldArgs& a = a_in;
  const int tiles = (a.M + TM - 1) / TM;
  const size_t lds = FIELD_LDS_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> opted{0};
  if (hipError_t e = ensure_dynamic_lds(opted, {reinterpret_cast<const void*>(k_field_fwd<false>), reinterpret_cast<const void*>(k_field_fwd<true>)}, (int)lds)) return e;
  if (save)
    hipLaunchKernelGGL(k_field_fwd<true>, dim3(tiles), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(k_field_fwd<false>, dim3(tiles), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace nerf
