// kernels.h -- kernel argument blocks and launchers shared by the .hip translation units and api.cpp.
#pragma once
#include "common.h"

namespace nerf {

constexpr int LDA = 260;  // floats per LDS activation row of the field kernels
constexpr int FIELD_LDS_FLOATS = TM * LDA + 4 * TM * 3;

struct FieldArgs {
  const float4* wp;        // packed weights (PACKED_FWD_F4 float4)
  Weights24 w;             // raw parameter pointers (biases, sigma / colour heads)
  const float* rayf;       // [B][RAYF]
  const float* dvec;       // [B][128]  b_dir + W_dir[:, :24] * gamma_dir(ray)
  const float* t;          // [M] depths
  float* rgb;              // [M][3]
  float* sigma;            // [M]
  float* pts_dbg;          // [M][3] or null
  float* gp_dbg;           // [M][60] or null
  float* save;             // training: [10][M][256] (h0..h7, feat, c(128 used)) or null
  float* spre;             // training: [M] sigma pre-activation
  int N;                   // samples per ray
  int M;                   // total samples (B*N)
};

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
  float* t_c;        // [B][Nc] or null
  float* d_cam;      // [B][3] or null
  float* d_wrd;      // [B][3] or null
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
};

struct MergeArgs {
  const float *t_c, *t_f, *sig_c, *sig_f, *rgb_c, *rgb_f;
  int B, Nc, Nf, P;
  float last;
  float* bundle;   // [B][N][5] or null
  float* w;        // [B][N] or null
  uint16_t* perm;  // [B][5][N] or null: sorted position -> original index (coarse first, then fine)
  float* C_fine;   // [B][3]
};

hipError_t launch_pack_weights(const Weights24& w, float4* out, int nseg, hipStream_t st);
hipError_t launch_field_fwd(const FieldArgs& a, bool save, hipStream_t st);
hipError_t launch_rays(const RaysArgs& a, hipStream_t st);
hipError_t launch_coarse(const CoarseArgs& a, hipStream_t st);
size_t merge_lds_bytes(int P);
hipError_t launch_merge(const MergeArgs& a, hipStream_t st);
hipError_t launch_ray_loss(const float* Cc, const float* Cf, const float* Ct, int B, float* loss, float* dCc, float* dCf, hipStream_t st);

}  // namespace nerf
