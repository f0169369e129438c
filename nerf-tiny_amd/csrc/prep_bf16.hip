// prep_bf16.hip -- everything a bf16-MLP call needs before its first field kernel, in ONE launch (MI355X / gfx950):
//   * the fold W_fold = W_dir[:, 24:] W_pi, b_fold (fold_block, 129 blocks -- the FIRST blocks of the grid),
//   * the packed weight image(s): the 32x32x16 forward stream + its bias block (training forward) or the 16x16x32 stream (inference),
//     and for a training call also the transposed stream of the backward chain (so nerf_hip_backward packs nothing),
//   * the ray records and coarse depths (ray_block: what k_rays does).
// As separate launches these were 3 (inference) or 4 (training: fold, pack, rays, and the backward's pack) dependent kernels of 5-8 us
// each -- 21-28 us of a 512-ray step that takes 143 / 740 us (forward / train step).  The only dependency among them is fold -> the
// 37 blocks that pack folded fragments (dir_info's h7 columns, forward and transposed, and the bias tiles that carry b_fold); those blocks
// wait for the fold blocks INSIDE the launch:
//   * every fold block, after its stores have drained (all threads: s_waitcnt vmcnt(0); barrier) releases at agent scope and stores this
//     call's TOKEN (handed in by the host, unique per call) into its own word of a 129-word array: a word left over from an earlier
//     call -- or never initialised -- cannot be mistaken for this call's, so nothing has to be zeroed between calls, and nobody does a
//     read-modify-write;
//   * a packing block that needs the fold polls the 129 words with 129 threads (relaxed, agent scope), goes through a barrier, ONE lane
//     acquires at agent scope (this CU's L1 may hold stale lines of the fold scratch from the previous call), barrier.
// Forward progress: fold blocks wait for nothing and are the LOWEST block indices of the grid.  The gfx950 workgroup dispatcher hands out the
// blocks of one launch in index order, so whenever a packing block is resident every fold block has been dispatched before it (it is
// resident or finished), and resident waves keep running whatever other streams (an overlap side stream, RCCL) hold of the chip.  HIP itself
// does not promise that order, so the wait is BOUNDED and a timeout is LOUD rather than silent:
//   * after `spin_limit` polls (2^22 x ~0.5 us: seconds, never a hang) the block gives up, sets NERF_HIP_STATUS_PREP_TIMEOUT in the sticky
//     status word, stores this call's token into the "timed-out" word (nerf_hip_read_status compares it with the "last packing call" word
//     block 0 stores: the flag stays up for every call that reuses the image with NERF_HIP_WEIGHTS_UNCHANGED), and
//   * POISONS what it packs: every fragment / bias tile that needed the fold is written as bf16 / fp32 NaN, so that the call's C_coarse,
//     C_fine, loss and gradients come out NaN -- a C-ABI caller that never reads a status word still cannot mistake them for pixels.
// NERF_PREP_BF16=0 selects the separate launches (no in-launch wait at all).  The per-call token is a kernel ARGUMENT: a captured stream
// must not be replayed (include/nerf_hip.h says so).  NERF_PREP_FAULT_INJECT=1 (tests): the fold blocks publish a wrong token and the
// waiters' bound is 2^10 polls, which exercises the whole timeout path on a healthy GPU.
#include <cstdlib>

#include "bf16_stream.h"
#include "bf16_weights.h"
#include "prep_parts.h"

namespace nerf {

constexpr int PREP_FOLD_BLOCKS = HALF + 1;
constexpr unsigned PREP_SPIN_LIMIT = 1u << 22;  // x ~0.5 us per poll: seconds, not forever
constexpr unsigned PREP_SPIN_LIMIT_INJECT = 1u << 10;
constexpr unsigned PREP_POISON_BF16X2 = 0x7fc07fc0u;  // two bf16 quiet NaNs
constexpr int PREP_TIMED_OUT_WORD = 1, PREP_PACKED_WORD = 2;  // relative to the sticky word (common.h STATUS_STICKY_WORD + 1 / + 2)
constexpr int PF_BLOCKS = (BF_NFRAG * 64 + 255) / 256;                       // 32x32x16 forward stream: 4 fragments per block
constexpr int PX_BLOCKS = (BX_NCHUNK * BF_CHUNK * 64 + 255) / 256;           // 16x16x32 inference stream
constexpr int PB_BLOCKS = (BBF_NCHUNK * BF_CHUNK * 64 + 255) / 256;          // transposed backward stream
constexpr int PBIAS_BLOCKS = BF_BIAS_BYTES / 4 / 256;                        // 16

struct PrepArgs {
  Weights24 w;
  float* fold;                 // FOLD_FLOATS scratch in the workspace
  unsigned char* img_fwd;      // forward image (bias block + stream) or null
  int fwd_form;                // 0: 32x32x16 (training / FORCE_TILE), 1: 16x16x32 (inference)
  unsigned char* img_bwd;      // transposed image or null
  unsigned* ready;             // [PREP_FOLD_BLOCKS]: fold block o stores this call's token into ready[o] when its row is out
  unsigned token;
  unsigned* sticky;            // status word that no kernel clears; sticky[1] = token of the last call whose wait timed out, sticky[2] = token of the last packing call
  unsigned spin_limit;         // polls before a waiting block gives up
  unsigned publish_xor;        // 0; fault injection: the fold blocks publish token ^ publish_xor (nobody's token)
  RaysArgs rays;               // rays.B = 0: no ray part
  int b_fwd0, b_bwd0, b_bias0, b_rays0, b_end;  // first block of every part (fold blocks first)
};

__device__ __forceinline__ void prep_publish_fold(const PrepArgs& a, const int block) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // one word per fold block, no read-modify-write: 129 blocks bumping ONE counter by compare-and-swap (the first version) serialised
    // into 0.2 ms of retries
    __hip_atomic_store(a.ready + block, a.token ^ a.publish_xor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// -> true: the fold is out and visible; false (block-uniform): the wait timed out -- the caller packs NaN instead of stale numbers
__device__ __forceinline__ bool prep_wait_fold(const PrepArgs& a) {
  int timed_out = 0;
  if (threadIdx.x < PREP_FOLD_BLOCKS) {  // thread t polls fold block t's word (relaxed, agent scope: an sc1 load, never this CU's L1)
    unsigned spins = 0;
    while (__hip_atomic_load(a.ready + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.token) {
      if (++spins >= a.spin_limit) {
        if (a.sticky) {
          atomicOr(a.sticky, 2u);  // NERF_HIP_STATUS_PREP_TIMEOUT
          __hip_atomic_store(a.sticky + PREP_TIMED_OUT_WORD, a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  const bool bad = __syncthreads_or(timed_out) != 0;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return !bad;
}

__global__ __launch_bounds__(256) void k_prep_bf16(const PrepArgs a) {
  const int b = blockIdx.x;
  const u32x4 poison = {PREP_POISON_BF16X2, PREP_POISON_BF16X2, PREP_POISON_BF16X2, PREP_POISON_BF16X2};
  if (b < a.b_fwd0) {  // (no fold blocks when nothing is packed: b_fwd0 = 0)
    // this launch packs a weight image: the word nerf_hip_read_status compares the "timed-out" word with
    if (b == 0 && threadIdx.x == 0 && a.sticky) __hip_atomic_store(a.sticky + PREP_PACKED_WORD, a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fold_block(a.w, a.fold, b);
    prep_publish_fold(a, b);
    return;
  }
  if (b < a.b_bwd0) {  // forward stream: 4 fragments per block
    const int gid = (b - a.b_fwd0) * 256 + threadIdx.x;
    const int frag = gid >> 6, lane = gid & 63;
    if (a.fwd_form == 0) {
      const int f0 = (gid - threadIdx.x) >> 6;  // block-uniform
      bool ok = true;
      if (f0 + 3 >= BFS_DIR && f0 < BFS_COL) ok = prep_wait_fold(a);
      if (frag >= BF_NFRAG) return;
      const int i = lane & 31, h = lane >> 5;
      u32x4 v = poison;
      if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
          v[q] = pack2(bf_weight(a.w, a.fold, frag, i, kk, h), bf_weight(a.w, a.fold, frag, i, kk + 1, h));
        }
      }
      *reinterpret_cast<u32x4*>(a.img_fwd + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
    } else {
      const int f0 = (gid - threadIdx.x) >> 6;
      bool ok = true;
      if (f0 + 3 >= BXS_DIR && f0 < BXS_COL) ok = prep_wait_fold(a);
      if (frag >= BX_NCHUNK * BF_CHUNK) return;
      const int i = lane & 15, q = lane >> 4;
      u32x4 v = poison;
      if (ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = 16 * (e >> 1) + 4 * q + 2 * (e & 1);
          v[e] = pack2(bx_weight(a.w, a.fold, frag, i, kk), bx_weight(a.w, a.fold, frag, i, kk + 1));
        }
      }
      *reinterpret_cast<u32x4*>(a.img_fwd + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
    }
    return;
  }
  if (b < a.b_bias0) {  // transposed stream of the backward chain
    const int gid = (b - a.b_bwd0) * 256 + threadIdx.x;
    const int frag = gid >> 6, lane = gid & 63, f0 = (gid - threadIdx.x) >> 6;
    bool ok = true;
    if (f0 + 3 >= BBS_FOLDT && f0 < BBS_L7T) ok = prep_wait_fold(a);
    if (frag >= BBF_NCHUNK * BF_CHUNK) return;
    const int i = lane & 31, h = lane >> 5;
    u32x4 v = poison;
    if (ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = 4 * h + 2 * (q & 1) + 8 * (q >> 1);
        v[q] = pack2(bb_weight(a.w, a.fold, frag, i, kk), bb_weight(a.w, a.fold, frag, i, kk + 1));
      }
    }
    *reinterpret_cast<u32x4*>(a.img_bwd + BF_BIAS_BYTES + (size_t)frag * BF_FRAG_BYTES + lane * 16) = v;
    return;
  }
  if (b < a.b_rays0) {  // bias block of the forward image (both forms share its float layout): 256 floats per block
    const int x = (b - a.b_bias0) * 256 + threadIdx.x;
    const int t0 = ((b - a.b_bias0) * 256) >> 5;  // first tile of this block (block-uniform); tiles BFB_DIR .. BFB_COL-1 carry b_fold
    bool ok = true;
    if (t0 + 7 >= BFB_DIR && t0 < BFB_COL) ok = prep_wait_fold(a);
    const int tile = x >> 5, i = x & 31;
    reinterpret_cast<float*>(a.img_fwd)[x] = !ok ? __uint_as_float(0x7fc00000u) : tile < BF_NBIAS_TILES ? bf_bias(a.w, a.fold, tile, i) : 0.f;
    return;
  }
  // ray records and coarse depths: two rays per block (the bf16 kernels take no per-ray start vector: rays.dvec is null, no barrier inside)
  const int ray = (b - a.b_rays0) * 2 + (threadIdx.x >> 7);
  if (ray < a.rays.B) ray_block(a.rays, ray, threadIdx.x & 127, nullptr);
}

// img_fwd: forward image of the chosen form (or null: weights unchanged); img_bwd: transposed image or null; rays.B = 0: no ray part
hipError_t launch_prep_bf16(const Weights24& w, float* fold, unsigned char* img_fwd, int fwd_form, unsigned char* img_bwd,
                            unsigned* ready, unsigned token, unsigned* sticky, const RaysArgs& rays, hipStream_t st) {
  if (rays.dvec) return hipErrorInvalidValue;  // (the fp32 path's per-ray start vector needs b_fold: it keeps k_rays)
  PrepArgs a;
  a.w = w; a.fold = fold; a.img_fwd = img_fwd; a.fwd_form = fwd_form; a.img_bwd = img_bwd; a.ready = ready; a.token = token; a.sticky = sticky;
  a.rays = rays;
  const char* inj = getenv("NERF_PREP_FAULT_INJECT");  // tests only: drive the timeout path (read per call so that a test can switch it)
  const bool inject = inj && atoi(inj) != 0;
  a.spin_limit = inject ? PREP_SPIN_LIMIT_INJECT : PREP_SPIN_LIMIT;
  a.publish_xor = inject ? 0x80000000u : 0u;
  int b = (img_fwd || img_bwd) ? PREP_FOLD_BLOCKS : 0;
  a.b_fwd0 = b; b += img_fwd ? (fwd_form == 0 ? PF_BLOCKS : PX_BLOCKS) : 0;
  a.b_bwd0 = b; b += img_bwd ? PB_BLOCKS : 0;
  a.b_bias0 = b; b += img_fwd ? PBIAS_BLOCKS : 0;
  a.b_rays0 = b; b += (rays.B + 1) / 2;
  a.b_end = b;
  hipLaunchKernelGGL(k_prep_bf16, dim3(b), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace nerf
