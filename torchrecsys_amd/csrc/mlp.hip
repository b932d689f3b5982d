// mlp.hip — the non-GEMM kernels of the MLP scorer (reference collaborative/mlp.py:88-115 and its autograd):
// embedding gather-concat, BatchNorm1d batch statistics (train mode, per scoring pass), BN+ReLU forward/backward,
// output layer (H -> 1) forward/backward, column reductions for bias / BN-affine gradients.
//
// Activations are stored for both scoring passes stacked: rows [0,B) = positive pass, rows [B,2B) = negative pass;
// BatchNorm statistics are taken per pass (the reference calls net.forward twice, model.py:171-185, so each call
// normalises over its own B rows and updates the running statistics once — SURVEY App. A.4).
#include "trs_common.h"

namespace {

constexpr int CHUNK_ROWS = 256;  // rows per partial-reduction chunk

// ------------------------------------------------------------------------------------------- gather-concat
struct GatherArgs {
  trs_tables T;
  trs_batch Bt;
  float* x;
  int64_t ld;
  int passes;
};

// one (row, field) segment of D floats per G-lane group; fields: 0 user, 1 item, 2+m metadata m
__global__ __launch_bounds__(TRS_BLOCK) void mlp_gather_kernel(const GatherArgs a) {
  const trs_tables& T = a.T;
  const int D = T.D, F = 2 + T.M;
  const int64_t B = a.Bt.B;
  const int ib = a.Bt.idx_bytes;
  const int64_t nseg = (int64_t)a.passes * B * F;
  const int chunks = (D + 3) / 4;
  const bool vec = (D % 4) == 0;
  const int64_t total = nseg * chunks;
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; e < total; e += stride) {
    const int64_t seg = e / chunks;
    const int c = (int)(e - seg * chunks);
    const int64_t row = seg / F;
    const int f = (int)(seg - row * F);
    const int pass = row >= B;
    const int64_t t = pass ? row - B : row;
    const float* tab;
    int64_t id, n_rows;
    if (f == 0) { tab = T.user; id = trs_ld_idx(a.Bt.user, ib, t); n_rows = T.n_users; }
    else if (f == 1) { tab = T.item; id = trs_ld_idx(pass ? a.Bt.neg : a.Bt.pos, ib, t); n_rows = T.n_items; }
    else {
      const int m = f - 2;
      tab = T.meta[m];
      id = trs_ld_idx(pass ? a.Bt.neg_meta : a.Bt.pos_meta, ib, t * T.M + m);
      n_rows = T.n_meta[m];
    }
    float* dst = a.x + row * a.ld + (int64_t)f * D + 4 * c;
    if ((uint64_t)id >= (uint64_t)n_rows) {
      if (c == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
      for (int q = 0; q < 4 && 4 * c + q < D; ++q) dst[q] = 0.f;
      continue;
    }
    const float* src = tab + id * (int64_t)D + 4 * c;
    if (vec && ((a.ld & 3) == 0)) {
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
      for (int q = 0; q < 4 && 4 * c + q < D; ++q) dst[q] = src[q];
    }
  }
}

// ------------------------------------------------------------------------------------------- BN statistics
// Partial statistics of one chunk of rows for every column: two passes over the chunk (the second one hits L2), so
// the partial M2 is taken around the chunk's own mean (no E[y^2]-E[y]^2 cancellation).
// part layout: (passes, n_chunks, 2, H): [.,.,0,:] = chunk mean, [.,.,1,:] = chunk M2.
__global__ __launch_bounds__(TRS_BLOCK) void bn_stats_partial_kernel(const float* __restrict__ y, int64_t rows_per_pass,
                                                                    int H, int64_t ld, int n_chunks,
                                                                    float* __restrict__ part) {
  const int col = blockIdx.x * TRS_BLOCK + threadIdx.x;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  if (col >= H) return;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < rows_per_pass) ? r0 + CHUNK_ROWS : rows_per_pass;
  const float* p = y + ((int64_t)pass * rows_per_pass) * ld + col;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += p[r * ld];
  const float mean = s / (float)(r1 - r0);
  float m2 = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float d = p[r * ld] - mean;
    m2 += d * d;
  }
  float* o = part + (((int64_t)pass * n_chunks + chunk) * 2) * H;
  o[col] = mean;
  o[H + col] = m2;
}

// Chan et al. pairwise combination of the chunk partials in fp64 -> batch mean, biased variance; running statistics
// update (momentum 0.1, unbiased variance) applied once per pass, positive pass first.
// Block = 64 columns x 4 segments of the chunk list; the 4 partial (n, mean, M2) triples are merged in LDS.
constexpr int FIN_COLS = 64, FIN_SEGS = TRS_BLOCK / FIN_COLS;

__device__ __forceinline__ void chan_merge(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
  if (nb == 0.0) return;
  const double tot = n + nb, delta = mb - mean;
  mean += delta * nb / tot;
  m2 += m2b + delta * delta * n * nb / tot;
  n = tot;
}

__global__ __launch_bounds__(TRS_BLOCK) void bn_stats_final_kernel(const float* __restrict__ part,
                                                                  int64_t rows_per_pass, int chunk_rows, int H,
                                                                  int n_chunks, int passes, float momentum,
                                                                  float* __restrict__ mean_out,
                                                                  float* __restrict__ var_out,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var) {
  __shared__ double s_n[FIN_SEGS][FIN_COLS], s_mean[FIN_SEGS][FIN_COLS], s_m2[FIN_SEGS][FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  for (int pass = 0; pass < passes; ++pass) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
    if (col < H) {
      for (int c = seg; c < n_chunks; c += FIN_SEGS) {
        const int64_t r0 = (int64_t)c * chunk_rows;
        const double nb = (double)((r0 + chunk_rows < rows_per_pass ? r0 + chunk_rows : rows_per_pass) - r0);
        const float* o = part + (((int64_t)pass * n_chunks + c) * 2) * H;
        chan_merge(n, mean, m2, nb, (double)o[col], (double)o[H + col]);
      }
    }
    s_n[seg][cl] = n; s_mean[seg][cl] = mean; s_m2[seg][cl] = m2;
    __syncthreads();
    if (seg == 0 && col < H) {
      for (int q = 1; q < FIN_SEGS; ++q) chan_merge(n, mean, m2, s_n[q][cl], s_mean[q][cl], s_m2[q][cl]);
      const float mu = (float)mean, var = (float)(m2 / n);
      mean_out[pass * H + col] = mu;
      var_out[pass * H + col] = var;
      if (running_mean) {
        const float unb = (float)(m2 / (n > 1.0 ? n - 1.0 : 1.0));
        running_mean[col] = (1.0f - momentum) * running_mean[col] + momentum * mu;
        running_var[col] = (1.0f - momentum) * running_var[col] + momentum * unb;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------- BN + ReLU forward
// out = relu(((y - mean) * invstd) * gamma + beta), invstd = 1/sqrt(var + eps); stats indexed per pass
// (stat_passes = 2: batch statistics; 1: running statistics in eval mode).  use_bn = 0: out = relu(y).
struct BnFwdArgs {
  const float* y;
  float* out;
  int64_t rows_per_pass, ld, ldo;
  int H, passes, stat_passes, use_bn;
  const float *mean, *var, *gamma, *beta;
  float eps;
};

__global__ __launch_bounds__(TRS_BLOCK) void bn_relu_fwd_kernel(const BnFwdArgs a) {
  const int H4 = (a.H + 3) / 4;
  const int64_t rows = a.rows_per_pass * a.passes;
  const int64_t total = rows * H4;
  const bool vec = (a.H % 4 == 0) && (a.ld % 4 == 0) && (a.ldo % 4 == 0);
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; e < total; e += stride) {
    const int64_t row = e / H4;
    const int c4 = (int)(e - row * H4) * 4;
    const int sp = (a.stat_passes > 1 && row >= a.rows_per_pass) ? 1 : 0;
    float v[4];
    const float* src = a.y + row * a.ld + c4;
    const int nq = (a.H - c4) < 4 ? (a.H - c4) : 4;
    if (vec) {
      const float4 t = *reinterpret_cast<const float4*>(src);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int q = 0; q < nq; ++q) v[q] = src[q];
    }
    for (int q = 0; q < nq; ++q) {
      float x = v[q];
      if (a.use_bn) {
        const int col = c4 + q;
        const float invstd = 1.0f / sqrtf(a.var[sp * a.H + col] + a.eps);
        x = ((x - a.mean[sp * a.H + col]) * invstd) * a.gamma[col] + a.beta[col];
      }
      v[q] = fmaxf(x, 0.f);
    }
    float* dst = a.out + row * a.ldo + c4;
    if (vec) {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int q = 0; q < nq; ++q) dst[q] = v[q];
    }
  }
}

// ------------------------------------------------------------------------------------------- BN + ReLU backward
// Given dx = dL/d(relu output), recompute yhat from y and the batch statistics:
//   dyhat = dx * [yhat > 0];  s1 = sum_rows dyhat;  s2 = sum_rows dyhat * xhat       (per pass, per column)
//   dy    = gamma * invstd * (dyhat - s1/B - xhat * s2/B)                            (train mode)
// Partial sums per chunk: part (passes, n_chunks, 2, H).  use_bn = 0: dy = dx * [y > 0], s1 = column sum of dy.
struct BnBwdArgs {
  const float* y;
  const float* dx;
  float* dy;
  int64_t rows_per_pass, ld, ldd;
  int H, passes, use_bn, n_chunks;
  const float *mean, *var, *gamma, *beta;
  float eps;
  float* part;
  const float* sums;  // (passes, 2, H) final s1, s2 (apply kernel)
};

__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
  const int col = blockIdx.x * TRS_BLOCK + threadIdx.x;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  if (col >= a.H) return;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < a.rows_per_pass) ? r0 + CHUNK_ROWS : a.rows_per_pass;
  const int64_t base = (int64_t)pass * a.rows_per_pass;
  float mu = 0.f, invstd = 1.f, ga = 1.f, be = 0.f;
  if (a.use_bn) {
    mu = a.mean[pass * a.H + col];
    invstd = 1.0f / sqrtf(a.var[pass * a.H + col] + a.eps);
    ga = a.gamma[col];
    be = a.beta[col];
  }
  float s1 = 0.f, s2 = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float yv = a.y[(base + r) * a.ld + col];
    const float xhat = (yv - mu) * invstd;
    const float yhat = a.use_bn ? xhat * ga + be : yv;
    const float d = yhat > 0.f ? a.dx[(base + r) * a.ldd + col] : 0.f;
    s1 += d;
    s2 += d * xhat;
  }
  float* o = a.part + (((int64_t)pass * a.n_chunks + chunk) * 2) * a.H;
  o[col] = s1;
  o[a.H + col] = s2;
}

// sums (passes,2,H) = sum over chunks (fp64 accumulate); dgamma = sum over passes of s2, dbeta = of s1.
// Block = 64 columns x 4 chunk-segments, merged in LDS in segment order (fixed order: reproducible).
__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_final_kernel(const float* __restrict__ part, int H, int n_chunks,
                                                                int passes, float* __restrict__ sums,
                                                                float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  __shared__ double s_1[FIN_SEGS][FIN_COLS], s_2[FIN_SEGS][FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  double g = 0.0, b = 0.0;
  for (int pass = 0; pass < passes; ++pass) {
    double s1 = 0.0, s2 = 0.0;
    if (col < H) {
      for (int c = seg; c < n_chunks; c += FIN_SEGS) {
        const float* o = part + (((int64_t)pass * n_chunks + c) * 2) * H;
        s1 += o[col];
        s2 += o[H + col];
      }
    }
    s_1[seg][cl] = s1; s_2[seg][cl] = s2;
    __syncthreads();
    if (seg == 0 && col < H) {
      for (int q = 1; q < FIN_SEGS; ++q) { s1 += s_1[q][cl]; s2 += s_2[q][cl]; }
      sums[(pass * 2 + 0) * H + col] = (float)s1;
      sums[(pass * 2 + 1) * H + col] = (float)s2;
      b += s1;
      g += s2;
    }
    __syncthreads();
  }
  if (seg == 0 && col < H) {
    if (dgamma) dgamma[col] = (float)g;
    if (dbeta) dbeta[col] = (float)b;
  }
}

__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  const int64_t rows = a.rows_per_pass * a.passes;
  const int64_t total = rows * a.H;
  const float invB = 1.0f / (float)a.rows_per_pass;
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; e < total; e += stride) {
    const int64_t row = e / a.H;
    const int col = (int)(e - row * a.H);
    const int pass = row >= a.rows_per_pass ? 1 : 0;
    const float yv = a.y[row * a.ld + col];
    const float dxv = a.dx[row * a.ldd + col];
    float out;
    if (a.use_bn) {
      const float mu = a.mean[pass * a.H + col];
      const float invstd = 1.0f / sqrtf(a.var[pass * a.H + col] + a.eps);
      const float ga = a.gamma[col];
      const float xhat = (yv - mu) * invstd;
      const float yhat = xhat * ga + a.beta[col];
      const float d = yhat > 0.f ? dxv : 0.f;
      const float s1 = a.sums[(pass * 2 + 0) * a.H + col], s2 = a.sums[(pass * 2 + 1) * a.H + col];
      out = (ga * invstd) * (d - s1 * invB - xhat * (s2 * invB));
    } else {
      out = yv > 0.f ? dxv : 0.f;
    }
    a.dy[row * a.ldd + col] = out;
  }
}

// ------------------------------------------------------------------------------------------- column sums
// part (passes, n_chunks, H) = per-chunk column sums of x (passes*rows_per_pass, H) [optionally weighted by w (rows)].
// Chunks never straddle a pass, and both passes are cut identically: when the negative pass's column is the exact
// negation of the positive pass's (hinge gradients -a/B, +a/B through an always-active unit) the two partial-sum
// sequences round identically and cancel to an exact 0, as the reference's two separate backward passes do.
__global__ __launch_bounds__(TRS_BLOCK) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows_per_pass,
                                                                  int H, int64_t ld, const float* __restrict__ w,
                                                                  int n_chunks, float* __restrict__ part) {
  const int col = blockIdx.x * TRS_BLOCK + threadIdx.x;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  if (col >= H) return;
  const int64_t base = (int64_t)pass * rows_per_pass;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < rows_per_pass) ? r0 + CHUNK_ROWS : rows_per_pass;
  float s = 0.f;
  if (w) {
    for (int64_t r = r0; r < r1; ++r) s += w[base + r] * x[(base + r) * ld + col];
  } else {
    for (int64_t r = r0; r < r1; ++r) s += x[(base + r) * ld + col];
  }
  part[((int64_t)pass * n_chunks + chunk) * H + col] = s;
}

__global__ __launch_bounds__(TRS_BLOCK) void colsum_final_kernel(const float* __restrict__ part, int H, int n_chunks,
                                                                int passes, float* __restrict__ out) {
  __shared__ double s_s[FIN_SEGS][FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  double tot = 0.0;
  for (int p = 0; p < passes; ++p) {  // per pass first: identical summation structure in both passes (exact cancellation)
    double s = 0.0;
    if (col < H)
      for (int c = seg; c < n_chunks; c += FIN_SEGS) s += part[((int64_t)p * n_chunks + c) * H + col];
    s_s[seg][cl] = s;
    __syncthreads();
    if (seg == 0) {
      for (int q = 1; q < FIN_SEGS; ++q) s += s_s[q][cl];
      tot += s;
    }
    __syncthreads();
  }
  if (seg == 0 && col < H) out[col] = (float)tot;
}

// ------------------------------------------------------------------------------------------- output layer H -> 1
// score[r] = sum_h x[r][h] * w[h] + b    (one wave per row; x rows are re-read by nobody: stream once)
__global__ __launch_bounds__(TRS_BLOCK) void rowdot_kernel(const float* __restrict__ x, int64_t rows, int H, int64_t ld,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  for (int64_t r = wave; r < rows; r += nwave) {
    float s = 0.f;
    for (int h = lane; h < H; h += 64) s += x[r * ld + h] * w[h];
    s = trs_wave_sum(s);
    if (lane == 0) out[r] = s + (b ? b[0] : 0.f);
  }
}

// dx[r][h] = g[r] * w[h]
__global__ __launch_bounds__(TRS_BLOCK) void outer_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                         int64_t rows, int H, float* __restrict__ dx, int64_t ld) {
  const int64_t total = rows * H;
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / H;
    const int h = (int)(e - r * H);
    dx[r * ld + h] = g[r] * w[h];
  }
}

static int n_chunks_of(int64_t rows) { return (int)((rows + CHUNK_ROWS - 1) / CHUNK_ROWS); }

}  // namespace

extern "C" int trs_mlp_gather_concat(const trs_tables* tables, const trs_batch* batch, int32_t passes, float* x_dev,
                                     int64_t ld, void* stream) {
  TRS_REQUIRE(tables && batch && x_dev, "trs_mlp_gather_concat: NULL argument");
  TRS_REQUIRE(tables->M >= 0 && tables->M <= TRS_MAX_META, "trs_mlp_gather_concat: bad M");
  TRS_REQUIRE(tables->user && tables->item && tables->D > 0, "trs_mlp_gather_concat: bad tables");
  for (int m = 0; m < tables->M; ++m) TRS_REQUIRE(tables->meta[m], "trs_mlp_gather_concat: metadata table %d NULL", m);
  TRS_REQUIRE(batch->idx_bytes == 4 || batch->idx_bytes == 8, "trs_mlp_gather_concat: idx_bytes must be 4 or 8");
  TRS_REQUIRE(ld >= (int64_t)(2 + tables->M) * tables->D, "trs_mlp_gather_concat: ld too small");
  if (batch->B == 0) return TRS_OK;
  TRS_REQUIRE(passes == 1 || passes == 2, "trs_mlp_gather_concat: passes must be 1 or 2");
  TRS_REQUIRE(batch->user && batch->pos && (passes == 1 || batch->neg), "trs_mlp_gather_concat: ids are NULL");
  TRS_REQUIRE(tables->M == 0 || (batch->pos_meta && (passes == 1 || batch->neg_meta)),
              "trs_mlp_gather_concat: metadata ids NULL");
  GatherArgs a = {*tables, *batch, x_dev, ld, passes};
  const int64_t total = (int64_t)passes * batch->B * (2 + tables->M) * ((tables->D + 3) / 4);
  hipLaunchKernelGGL(mlp_gather_kernel, dim3(trs_grid(total, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream, a);
  TRS_CHECK_LAUNCH("mlp_gather_kernel");
  return TRS_OK;
}

extern "C" int64_t trs_bn_workspace_floats(int64_t rows_per_pass, int32_t H, int32_t passes) {
  return (int64_t)passes * n_chunks_of(rows_per_pass) * 2 * H;
}

extern "C" int trs_bn_batch_stats(const float* y_dev, int64_t rows_per_pass, int32_t H, int64_t ld, int32_t passes,
                                  float momentum, float* mean_out_dev, float* var_out_dev, float* running_mean_dev,
                                  float* running_var_dev, float* workspace_dev, void* stream) {
  TRS_REQUIRE(y_dev && mean_out_dev && var_out_dev && workspace_dev, "trs_bn_batch_stats: NULL argument");
  TRS_REQUIRE(rows_per_pass > 0 && H > 0 && ld >= H && passes >= 1 && passes <= 2, "trs_bn_batch_stats: bad shape");
  TRS_REQUIRE((running_mean_dev == nullptr) == (running_var_dev == nullptr), "trs_bn_batch_stats: running stats");
  const int nc = n_chunks_of(rows_per_pass);
  const int gx = (H + TRS_BLOCK - 1) / TRS_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, y_dev, rows_per_pass, H, ld,
                     nc, workspace_dev);
  TRS_CHECK_LAUNCH("bn_stats_partial_kernel");
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((H + FIN_COLS - 1) / FIN_COLS), dim3(TRS_BLOCK), 0, s, workspace_dev, rows_per_pass, CHUNK_ROWS, H, nc, passes,
                     momentum, mean_out_dev, var_out_dev, running_mean_dev, running_var_dev);
  TRS_CHECK_LAUNCH("bn_stats_final_kernel");
  return TRS_OK;
}

extern "C" int trs_bn_stats_finalize(const float* part_dev, int64_t rows_per_pass, int32_t chunk_rows, int32_t H,
                                     int32_t passes, float momentum, float* mean_out_dev, float* var_out_dev,
                                     float* running_mean_dev, float* running_var_dev, void* stream) {
  TRS_REQUIRE(part_dev && mean_out_dev && var_out_dev, "trs_bn_stats_finalize: NULL argument");
  TRS_REQUIRE(rows_per_pass > 0 && chunk_rows > 0 && H > 0 && passes >= 1 && passes <= 2,
              "trs_bn_stats_finalize: bad shape");
  TRS_REQUIRE(passes == 1 || rows_per_pass % chunk_rows == 0,
              "trs_bn_stats_finalize: a chunk must not straddle the two passes (rows_per_pass %% chunk_rows != 0)");
  TRS_REQUIRE((running_mean_dev == nullptr) == (running_var_dev == nullptr), "trs_bn_stats_finalize: running stats");
  const int nc = (int)((rows_per_pass + chunk_rows - 1) / chunk_rows);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((H + FIN_COLS - 1) / FIN_COLS), dim3(TRS_BLOCK), 0,
                     (hipStream_t)stream, part_dev, rows_per_pass, chunk_rows, H, nc, passes, momentum, mean_out_dev,
                     var_out_dev, running_mean_dev, running_var_dev);
  TRS_CHECK_LAUNCH("bn_stats_final_kernel");
  return TRS_OK;
}

extern "C" int trs_bn_relu_forward(const float* y_dev, int64_t rows_per_pass, int32_t passes, int32_t H, int64_t ld,
                                   int32_t use_bn, int32_t stat_passes, const float* mean_dev, const float* var_dev,
                                   const float* gamma_dev, const float* beta_dev, float eps, float* out_dev,
                                   int64_t ldo, void* stream) {
  TRS_REQUIRE(y_dev && out_dev, "trs_bn_relu_forward: NULL argument");
  TRS_REQUIRE(rows_per_pass >= 0 && H > 0 && ld >= H && ldo >= H && passes >= 1, "trs_bn_relu_forward: bad shape");
  TRS_REQUIRE(!use_bn || (mean_dev && var_dev && gamma_dev && beta_dev), "trs_bn_relu_forward: BN tensors are NULL");
  TRS_REQUIRE(stat_passes == 1 || stat_passes == passes, "trs_bn_relu_forward: stat_passes must be 1 or passes");
  if (rows_per_pass == 0) return TRS_OK;
  BnFwdArgs a = {y_dev, out_dev, rows_per_pass, ld, ldo, H, passes, stat_passes, use_bn, mean_dev, var_dev, gamma_dev,
                 beta_dev, eps};
  const int64_t total = rows_per_pass * passes * ((H + 3) / 4);
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(trs_grid(total, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream, a);
  TRS_CHECK_LAUNCH("bn_relu_fwd_kernel");
  return TRS_OK;
}

extern "C" int trs_bn_relu_backward(const float* y_dev, const float* dx_dev, int64_t rows_per_pass, int32_t passes,
                                    int32_t H, int64_t ld, int64_t ldd, int32_t use_bn, const float* mean_dev,
                                    const float* var_dev, const float* gamma_dev, const float* beta_dev, float eps,
                                    float* dy_dev, float* dgamma_dev, float* dbeta_dev, float* workspace_dev,
                                    void* stream) {
  TRS_REQUIRE(y_dev && dx_dev && dy_dev && workspace_dev, "trs_bn_relu_backward: NULL argument");
  TRS_REQUIRE(rows_per_pass > 0 && H > 0 && ld >= H && ldd >= H && passes >= 1 && passes <= 2,
              "trs_bn_relu_backward: bad shape");
  TRS_REQUIRE(!use_bn || (mean_dev && var_dev && gamma_dev && beta_dev), "trs_bn_relu_backward: BN tensors are NULL");
  const int nc = n_chunks_of(rows_per_pass);
  const int gx = (H + TRS_BLOCK - 1) / TRS_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  float* sums = workspace_dev + (int64_t)passes * nc * 2 * H;  // (passes,2,H) behind the partials
  BnBwdArgs a = {y_dev, dx_dev, dy_dev, rows_per_pass, ld, ldd, H, passes, use_bn, nc, mean_dev, var_dev, gamma_dev,
                 beta_dev, eps, workspace_dev, sums};
  if (use_bn) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, a);
    TRS_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((H + FIN_COLS - 1) / FIN_COLS), dim3(TRS_BLOCK), 0, s, workspace_dev, H, nc, passes, sums,
                       dgamma_dev, dbeta_dev);
    TRS_CHECK_LAUNCH("bn_bwd_final_kernel");
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(trs_grid(rows_per_pass * passes * H, TRS_BLOCK)), dim3(TRS_BLOCK), 0, s,
                     a);
  TRS_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return TRS_OK;
}

extern "C" int64_t trs_bn_backward_workspace_floats(int64_t rows_per_pass, int32_t H, int32_t passes) {
  return (int64_t)passes * n_chunks_of(rows_per_pass) * 2 * H + (int64_t)passes * 2 * H;
}

extern "C" int trs_colsum(const float* x_dev, int64_t rows_per_pass, int32_t passes, int32_t H, int64_t ld,
                          const float* row_weight_dev, float* out_dev, float* workspace_dev, void* stream) {
  TRS_REQUIRE(x_dev && out_dev && workspace_dev, "trs_colsum: NULL argument");
  TRS_REQUIRE(rows_per_pass > 0 && passes >= 1 && H > 0 && ld >= H, "trs_colsum: bad shape");
  const int nc = n_chunks_of(rows_per_pass);
  const int gx = (H + TRS_BLOCK - 1) / TRS_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, x_dev, rows_per_pass, H, ld,
                     row_weight_dev, nc, workspace_dev);
  TRS_CHECK_LAUNCH("colsum_partial_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((H + FIN_COLS - 1) / FIN_COLS), dim3(TRS_BLOCK), 0, s, workspace_dev, H, nc, passes, out_dev);
  TRS_CHECK_LAUNCH("colsum_final_kernel");
  return TRS_OK;
}

extern "C" int64_t trs_colsum_workspace_floats(int64_t rows_per_pass, int32_t passes, int32_t H) {
  return (int64_t)passes * n_chunks_of(rows_per_pass) * H;
}

extern "C" int trs_rowdot(const float* x_dev, int64_t rows, int32_t H, int64_t ld, const float* w_dev,
                          const float* bias_dev, float* out_dev, void* stream) {
  TRS_REQUIRE(x_dev && w_dev && out_dev, "trs_rowdot: NULL argument");
  TRS_REQUIRE(rows >= 0 && H > 0 && ld >= H, "trs_rowdot: bad shape");
  if (rows == 0) return TRS_OK;
  hipLaunchKernelGGL(rowdot_kernel, dim3(trs_grid(rows, TRS_BLOCK / TRS_WAVE)), dim3(TRS_BLOCK), 0, (hipStream_t)stream,
                     x_dev, rows, H, ld, w_dev, bias_dev, out_dev);
  TRS_CHECK_LAUNCH("rowdot_kernel");
  return TRS_OK;
}

extern "C" int trs_outer(const float* g_dev, const float* w_dev, int64_t rows, int32_t H, float* dx_dev, int64_t ld,
                         void* stream) {
  TRS_REQUIRE(g_dev && w_dev && dx_dev, "trs_outer: NULL argument");
  TRS_REQUIRE(rows >= 0 && H > 0 && ld >= H, "trs_outer: bad shape");
  if (rows == 0) return TRS_OK;
  hipLaunchKernelGGL(outer_kernel, dim3(trs_grid(rows * H, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream, g_dev,
                     w_dev, rows, H, dx_dev, ld);
  TRS_CHECK_LAUNCH("outer_kernel");
  return TRS_OK;
}
