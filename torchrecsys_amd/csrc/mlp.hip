// mlp.hip — the non-GEMM kernels of the MLP scorer (reference collaborative/mlp.py:88-115 and its autograd):
// embedding gather-concat, BatchNorm1d batch statistics (train mode, per scoring pass), BN+ReLU forward/backward,
// output layer (H -> 1) forward/backward, column reductions for bias / BN-affine gradients.
//
// Activations are stored for both scoring passes stacked: rows [0,B) = positive pass, rows [B,2B) = negative pass;
// BatchNorm statistics are taken per pass (the reference calls net.forward twice, model.py:171-185, so each call
// normalises over its own B rows and updates the running statistics once — SURVEY App. A.4).
#include "trs_common.h"

namespace {

constexpr int CHUNK_ROWS = 128;  // rows per partial-reduction chunk = per workgroup of the streaming kernels

// ------------------------------------------------------------------------------------------- gather-concat
__device__ __forceinline__ unsigned short f2bf(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// 4 floats -> 4 bf16 (RNE) as one 8-byte store
__device__ __forceinline__ void st_bf16x4(unsigned short* dst, float a, float b, float c, float d) {
  uint2 v;
  v.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  v.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  *reinterpret_cast<uint2*>(dst) = v;
}

struct GatherArgs {
  trs_tables T;
  trs_batch Bt;
  float* x;             // fp32 image, or NULL
  int64_t ld;
  int passes;
  unsigned short* x16;  // bf16 image (use_amp: what the bf16-resident GEMMs read), or NULL; same ld (elements)
};

// one (row, field) segment of D floats per lane group (a whole wave at D = 256), lane = 4-float chunk (fields: 0 user,
// 1 item, 2+m metadata m): the index arithmetic (a 64-bit division) is per segment, not per element.  Four segments per
// turn: their ids, then their rows, are loaded together (one row per turn left 16 KB in flight per CU: 3.8 TB/s).
__global__ __launch_bounds__(TRS_BLOCK) void mlp_gather_kernel(const GatherArgs a) {
  const trs_tables& T = a.T;
  const int D = T.D, F = 2 + T.M;
  const int64_t B = a.Bt.B;
  const int ib = a.Bt.idx_bytes;
  const int64_t nseg = (int64_t)a.passes * B * F;
  const int chunks = (D + 3) / 4;
  const bool vec = (D % 4) == 0 && (a.ld & 3) == 0;
  const int lane = threadIdx.x & 63;
  int gsh = 0;  // lane group of G = 2^gsh lanes per segment (the smallest power of two >= chunks, at most a wave)
  while (gsh < 6 && (1 << gsh) < chunks) ++gsh;
  const int G = 1 << gsh, spw = TRS_WAVE >> gsh, lig = lane & (G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  constexpr int U = 4;
  for (int64_t seg0 = wave * spw * U + (lane >> gsh); seg0 < nseg; seg0 += nwave * spw * U) {
    int64_t row[U], id[U];
    const float* tab[U];
    int f[U];
    bool bad[U], on[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t seg = seg0 + (int64_t)k * spw;
      on[k] = seg < nseg;
      const int64_t sc = on[k] ? seg : nseg - 1;
      row[k] = sc / F;
      f[k] = (int)(sc - row[k] * F);
      const int pass = row[k] >= B;
      const int64_t t = pass ? row[k] - B : row[k];
      int64_t n_rows;
      if (f[k] == 0) { tab[k] = T.user; id[k] = trs_ld_idx(a.Bt.user, ib, t); n_rows = T.n_users; }
      else if (f[k] == 1) { tab[k] = T.item; id[k] = trs_ld_idx(pass ? a.Bt.neg : a.Bt.pos, ib, t); n_rows = T.n_items; }
      else {
        const int m = f[k] - 2;
        tab[k] = T.meta[m];
        id[k] = trs_ld_idx(pass ? a.Bt.neg_meta : a.Bt.pos_meta, ib, t * T.M + m);
        n_rows = T.n_meta[m];
      }
      bad[k] = (uint64_t)id[k] >= (uint64_t)n_rows;
      if (bad[k]) id[k] = 0;
    }
    for (int c = lig; c < chunks; c += G) {
      if (vec) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = *reinterpret_cast<const float4*>(tab[k] + id[k] * (int64_t)D + 4 * c);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (!on[k]) continue;
          if (bad[k]) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          const int64_t doff = row[k] * a.ld + (int64_t)f[k] * D + 4 * c;
          if (a.x16) st_bf16x4(a.x16 + doff, v[k].x, v[k].y, v[k].z, v[k].w);
          if (a.x) *reinterpret_cast<float4*>(a.x + doff) = v[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (!on[k]) continue;
          const int64_t doff = row[k] * a.ld + (int64_t)f[k] * D + 4 * c;
          const float* src = tab[k] + id[k] * (int64_t)D + 4 * c;
          for (int q = 0; q < 4 && 4 * c + q < D; ++q) {
            const float val = bad[k] ? 0.f : src[q];
            if (a.x16) a.x16[doff + q] = f2bf(val);
            if (a.x) a.x[doff + q] = val;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (on[k] && bad[k] && lig == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
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
constexpr int FIN_COLS = 16, FIN_SEGS = TRS_BLOCK / FIN_COLS;  // narrow column slabs: H/16 workgroups, 16 segments each
constexpr int FIN_U = 8;  // chunk partials a thread has in flight

__global__ __launch_bounds__(TRS_BLOCK) void bn_stats_final_kernel(const float* __restrict__ part,
                                                                  int64_t rows_per_pass, int chunk_rows, int H,
                                                                  int n_chunks, float* __restrict__ mean_out,
                                                                  float* __restrict__ var_out) {
  // Chan et al. combination of the per-chunk (mean, M2) pairs in fp64, written as two weighted sums so that no
  // division sits in the loops: mean = sum(nb*mean_b)/N, then M2 = sum(M2_b + nb*(mean_b - mean)^2).
  __shared__ double s_a[FIN_SEGS][FIN_COLS], s_b[FIN_SEGS][FIN_COLS];
  __shared__ double s_mean[FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  const int pass = blockIdx.y;
  const float* base = part + (int64_t)pass * n_chunks * 2 * H;
  double s1 = 0.0, cnt = 0.0;
  // FIN_U partials in flight per turn (the loads are issued before the first add; summation order unchanged)
  if (col < H) {
    for (int c0 = seg; c0 < n_chunks; c0 += FIN_U * FIN_SEGS) {
      float v[FIN_U];
#pragma unroll
      for (int k = 0; k < FIN_U; ++k) {
        const int c = c0 + k * FIN_SEGS;
        v[k] = base[(int64_t)(c < n_chunks ? c : n_chunks - 1) * 2 * H + col];
      }
#pragma unroll
      for (int k = 0; k < FIN_U; ++k) {
        const int c = c0 + k * FIN_SEGS;
        const int64_t r0 = (int64_t)c * chunk_rows;
        const double nb = (double)((r0 + chunk_rows < rows_per_pass ? r0 + chunk_rows : rows_per_pass) - r0);
        if (c < n_chunks) {
          s1 += nb * (double)v[k];
          cnt += nb;
        }
      }
    }
  }
  s_a[seg][cl] = s1; s_b[seg][cl] = cnt;
  __syncthreads();
  if (seg == 0) {
    for (int q = 1; q < FIN_SEGS; ++q) { s1 += s_a[q][cl]; cnt += s_b[q][cl]; }
    s_mean[cl] = s1 / cnt;
  }
  __syncthreads();
  const double mean = s_mean[cl];
  const double n = (double)rows_per_pass;
  double m2 = 0.0;
  if (col < H) {
    for (int c0 = seg; c0 < n_chunks; c0 += FIN_U * FIN_SEGS) {
      float v[FIN_U], w[FIN_U];
#pragma unroll
      for (int k = 0; k < FIN_U; ++k) {
        const int c = c0 + k * FIN_SEGS;
        const float* o = base + (int64_t)(c < n_chunks ? c : n_chunks - 1) * 2 * H;
        v[k] = o[col];
        w[k] = o[H + col];
      }
#pragma unroll
      for (int k = 0; k < FIN_U; ++k) {
        const int c = c0 + k * FIN_SEGS;
        const int64_t r0 = (int64_t)c * chunk_rows;
        const double nb = (double)((r0 + chunk_rows < rows_per_pass ? r0 + chunk_rows : rows_per_pass) - r0);
        const double d = (double)v[k] - mean;
        if (c < n_chunks) m2 += (double)w[k] + nb * d * d;
      }
    }
  }
  __syncthreads();
  s_a[seg][cl] = m2;
  __syncthreads();
  if (seg == 0 && col < H) {
    for (int q = 1; q < FIN_SEGS; ++q) m2 += s_a[q][cl];
    const float mu = (float)mean, var = (float)(m2 / n);
    mean_out[pass * H + col] = mu;
    var_out[pass * H + col] = var;
  }
}

// The same finalise when a thread's share of the chunk list fits in registers (FIN_KEEP = 8 / 16 / 32 chunks per thread —
// c3: 32, c5: 16): every (mean, M2) partial is loaded ONCE, all loads issued before the first add; the second sum runs from
// registers.  Same order of additions as the two-sweep kernel above (bit-identical output), half the dependent round
// trips to L2 (TRS_BN_FINAL_TWO_SWEEPS=1 forces the two-sweep kernel: tests).
template <int FIN_KEEP>
__global__ __launch_bounds__(TRS_BLOCK) void bn_stats_final_regs_kernel(const float* __restrict__ part,
                                                                       int64_t rows_per_pass, int chunk_rows, int H,
                                                                       int n_chunks, float* __restrict__ mean_out,
                                                                       float* __restrict__ var_out) {
  __shared__ double s_a[FIN_SEGS][FIN_COLS], s_b[FIN_SEGS][FIN_COLS];
  __shared__ double s_mean[FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  const int pass = blockIdx.y;
  const float* base = part + (int64_t)pass * n_chunks * 2 * H;
  const int cc = col < H ? col : H - 1;
  float v[FIN_KEEP], w[FIN_KEEP];
#pragma unroll
  for (int k = 0; k < FIN_KEEP; ++k) {
    const int c = seg + k * FIN_SEGS;
    const float* o = base + (int64_t)(c < n_chunks ? c : n_chunks - 1) * 2 * H;
    v[k] = o[cc];
    w[k] = o[H + cc];
  }
  double s1 = 0.0, cnt = 0.0;
#pragma unroll
  for (int k = 0; k < FIN_KEEP; ++k) {
    const int c = seg + k * FIN_SEGS;
    const int64_t r0 = (int64_t)c * chunk_rows;
    const double nb = (double)((r0 + chunk_rows < rows_per_pass ? r0 + chunk_rows : rows_per_pass) - r0);
    if (c < n_chunks && col < H) {
      s1 += nb * (double)v[k];
      cnt += nb;
    }
  }
  s_a[seg][cl] = s1; s_b[seg][cl] = cnt;
  __syncthreads();
  if (seg == 0) {
    for (int q = 1; q < FIN_SEGS; ++q) { s1 += s_a[q][cl]; cnt += s_b[q][cl]; }
    s_mean[cl] = s1 / cnt;
  }
  __syncthreads();
  const double mean = s_mean[cl];
  const double n = (double)rows_per_pass;
  double m2 = 0.0;
#pragma unroll
  for (int k = 0; k < FIN_KEEP; ++k) {
    const int c = seg + k * FIN_SEGS;
    const int64_t r0 = (int64_t)c * chunk_rows;
    const double nb = (double)((r0 + chunk_rows < rows_per_pass ? r0 + chunk_rows : rows_per_pass) - r0);
    const double d = (double)v[k] - mean;
    if (c < n_chunks && col < H) m2 += (double)w[k] + nb * d * d;
  }
  __syncthreads();
  s_a[seg][cl] = m2;
  __syncthreads();
  if (seg == 0 && col < H) {
    for (int q = 1; q < FIN_SEGS; ++q) m2 += s_a[q][cl];
    const float mu = (float)mean, var = (float)(m2 / n);
    mean_out[pass * H + col] = mu;
    var_out[pass * H + col] = var;
  }
}

// finalise launcher: the register kernel when a thread's chunks fit, the two-sweep kernel otherwise
static void launch_bn_stats_final(const float* part, int64_t rows_per_pass, int chunk_rows, int H, int nc, int passes,
                                  float* mean_out, float* var_out, hipStream_t s) {
  const dim3 gr((H + FIN_COLS - 1) / FIN_COLS, passes), bl(TRS_BLOCK);
  const bool two_sweeps = trs_tuning().bn_final_two_sweeps != 0;
  const int per_thread = (nc + FIN_SEGS - 1) / FIN_SEGS;
  if (two_sweeps) {
    hipLaunchKernelGGL(bn_stats_final_kernel, gr, bl, 0, s, part, rows_per_pass, chunk_rows, H, nc, mean_out, var_out);
  } else if (per_thread <= 8) {
    hipLaunchKernelGGL(bn_stats_final_regs_kernel<8>, gr, bl, 0, s, part, rows_per_pass, chunk_rows, H, nc, mean_out, var_out);
  } else if (per_thread <= 16) {
    hipLaunchKernelGGL(bn_stats_final_regs_kernel<16>, gr, bl, 0, s, part, rows_per_pass, chunk_rows, H, nc, mean_out, var_out);
  } else if (per_thread <= 32) {
    hipLaunchKernelGGL(bn_stats_final_regs_kernel<32>, gr, bl, 0, s, part, rows_per_pass, chunk_rows, H, nc, mean_out, var_out);
  } else {
    hipLaunchKernelGGL(bn_stats_final_kernel, gr, bl, 0, s, part, rows_per_pass, chunk_rows, H, nc, mean_out, var_out);
  }
}

// running statistics: one momentum update per pass, in pass order (the reference's two net.forward calls, mlp.py:88-115)
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                         int64_t rows_per_pass, int H, int passes, float momentum,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         long long* tracked) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= H) return;
  if (col == 0 && tracked) *tracked += passes;
  const float n = (float)rows_per_pass;
  float rm = running_mean[col], rv = running_var[col];
  for (int pass = 0; pass < passes; ++pass) {
    const float mu = mean[pass * H + col];
    const double m2 = (double)var[pass * H + col] * (double)n;
    const float unb = (float)(m2 / (n > 1.f ? (double)n - 1.0 : 1.0));
    rm = (1.0f - momentum) * rm + momentum * mu;
    rv = (1.0f - momentum) * rv + momentum * unb;
  }
  running_mean[col] = rm;
  running_var[col] = rv;
}

// ------------------------------------------------------------------------------------------- BN + ReLU forward
// out = relu(((y - mean) * invstd) * gamma + beta), invstd = 1/sqrt(var + eps); stats indexed per pass
// (stat_passes = 2: batch statistics; 1: running statistics in eval mode).  use_bn = 0: out = relu(y).
struct BnFwdArgs {
  const float* y;
  float* out;             // fp32 output, or NULL
  unsigned short* out16;  // bf16 output (row stride ldo elements), or NULL
  int64_t rows_per_pass, ld, ldo;
  int H, passes, stat_passes, use_bn;
  const float *mean, *var, *gamma, *beta;
  float eps;
  const unsigned short* y16;  // bf16 image of y (bf16-resident path) when y is NULL
  // optional: the running-statistics momentum update of bn_running_update_kernel done by the first row block's threads
  // (train mode, stat_passes == passes); NULL = not here
  float momentum;
  float* running_mean;
  float* running_var;
  long long* tracked;  // BatchNorm1d.num_batches_tracked (+= passes with the running update), or NULL
  // optional (v4 kernel, H == 4 * threads-per-row <= 256: a row's columns sit in one wave): the H -> 1 output layer
  // dot_out[row] = sum_c out[row][c] * dot_w[c] + dot_b[0] formed from the registers that hold the row; NULL = not here
  const float* dot_w;
  const float* dot_b;
  float* dot_out;
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
  float* cs_part;     // optional (passes, n_chunks, H): per-chunk column sums of dy (the layer's bias gradient)
  unsigned short* dy16;  // optional bf16 copy of dy (row stride ldd elements): what the bf16-resident GEMMs read
  const unsigned short* y16;   // bf16 images of y / dx when the fp32 pointers are NULL (bf16-resident path)
  const unsigned short* dx16;
  float inv_n;  // 1 / rows the statistics were taken over (rows_per_pass; world * rows_per_pass under sync-BatchNorm)
  // OUTER kernels: dx[r][c] = og[r] * ow[c] is formed on the fly (the H -> 1 output layer's input gradient, never stored)
  const float* og;
  const float* ow;
  // OUTER reduce kernel, optional: (passes, n_chunks, H) per-chunk column sums of og[r] * relu(bn(y))[r][c] — the output
  // layer's weight gradient, from the activations the kernel recomputes anyway (they are then never stored)
  float* xw_part;
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
      for (int c0 = seg; c0 < n_chunks; c0 += FIN_U * FIN_SEGS) {
        float v[FIN_U], w[FIN_U];
#pragma unroll
        for (int k = 0; k < FIN_U; ++k) {
          const int c = c0 + k * FIN_SEGS;
          const float* o = part + (((int64_t)pass * n_chunks + (c < n_chunks ? c : n_chunks - 1)) * 2) * H;
          v[k] = o[col];
          w[k] = o[H + col];
        }
#pragma unroll
        for (int k = 0; k < FIN_U; ++k)
          if (c0 + k * FIN_SEGS < n_chunks) {
            s1 += v[k];
            s2 += w[k];
          }
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

// The same with both passes' partials of a thread loaded at once into registers (KEEP chunks per thread and pass: c5 16,
// c3 32): one round of loads instead of one per pass and 8-chunk turn; same order of additions (bit-identical).
template <int KEEP>
__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_final_regs_kernel(const float* __restrict__ part, int H, int n_chunks,
                                                                     int passes, float* __restrict__ sums,
                                                                     float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta) {
  __shared__ double s_1[2][FIN_SEGS][FIN_COLS], s_2[2][FIN_SEGS][FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  const int cc = col < H ? col : H - 1;
  float v[2][KEEP], w[2][KEEP];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
      const int c = seg + k * FIN_SEGS;
      const float* o = part + (((int64_t)(p < passes ? p : passes - 1) * n_chunks + (c < n_chunks ? c : n_chunks - 1)) * 2) * H;
      v[p][k] = o[cc];
      w[p][k] = o[H + cc];
    }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < KEEP; ++k)
      if (seg + k * FIN_SEGS < n_chunks && col < H) {
        s1 += v[p][k];
        s2 += w[p][k];
      }
    s_1[p][seg][cl] = s1; s_2[p][seg][cl] = s2;
  }
  __syncthreads();
  if (seg == 0 && col < H) {
    double g = 0.0, b = 0.0;
    for (int p = 0; p < passes; ++p) {
      double s1 = s_1[p][0][cl], s2 = s_2[p][0][cl];
      for (int q = 1; q < FIN_SEGS; ++q) { s1 += s_1[p][q][cl]; s2 += s_2[p][q][cl]; }
      sums[(p * 2 + 0) * H + col] = (float)s1;
      sums[(p * 2 + 1) * H + col] = (float)s2;
      b += s1;
      g += s2;
    }
    if (dgamma) dgamma[col] = (float)g;
    if (dbeta) dbeta[col] = (float)b;
  }
}

static void launch_bn_bwd_final(const float* part, int H, int nc, int passes, float* sums, float* dgamma, float* dbeta,
                                hipStream_t s) {
  const dim3 gr((H + FIN_COLS - 1) / FIN_COLS), bl(TRS_BLOCK);
  const bool two_sweeps = trs_tuning().bn_final_two_sweeps != 0;
  const int per_thread = (nc + FIN_SEGS - 1) / FIN_SEGS;
  if (two_sweeps || passes > 2 || per_thread > 32)
    hipLaunchKernelGGL(bn_bwd_final_kernel, gr, bl, 0, s, part, H, nc, passes, sums, dgamma, dbeta);
  else if (per_thread <= 16)
    hipLaunchKernelGGL(bn_bwd_final_regs_kernel<16>, gr, bl, 0, s, part, H, nc, passes, sums, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn_bwd_final_regs_kernel<32>, gr, bl, 0, s, part, H, nc, passes, sums, dgamma, dbeta);
}

__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  const int64_t rows = a.rows_per_pass * a.passes;
  const int64_t total = rows * a.H;
  const float invB = a.inv_n;
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
      for (int c0 = seg; c0 < n_chunks; c0 += FIN_U * FIN_SEGS) {
        float v[FIN_U];
#pragma unroll
        for (int k = 0; k < FIN_U; ++k) {
          const int c = c0 + k * FIN_SEGS;
          v[k] = part[((int64_t)p * n_chunks + (c < n_chunks ? c : n_chunks - 1)) * H + col];
        }
#pragma unroll
        for (int k = 0; k < FIN_U; ++k)
          if (c0 + k * FIN_SEGS < n_chunks) s += v[k];
      }
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

template <int KEEP>
__global__ __launch_bounds__(TRS_BLOCK) void colsum_final_regs_kernel(const float* __restrict__ part, int H, int n_chunks,
                                                                     int passes, float* __restrict__ out) {
  // both passes' partials of a thread in registers, one round of loads; additions in colsum_final_kernel's order
  __shared__ double s_s[2][FIN_SEGS][FIN_COLS];
  const int cl = threadIdx.x % FIN_COLS, seg = threadIdx.x / FIN_COLS;
  const int col = blockIdx.x * FIN_COLS + cl;
  const int cc = col < H ? col : H - 1;
  float v[2][KEEP];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
      const int c = seg + k * FIN_SEGS;
      v[p][k] = part[((int64_t)(p < passes ? p : passes - 1) * n_chunks + (c < n_chunks ? c : n_chunks - 1)) * H + cc];
    }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    double sm = 0.0;
#pragma unroll
    for (int k = 0; k < KEEP; ++k)
      if (seg + k * FIN_SEGS < n_chunks && col < H) sm += v[p][k];
    s_s[p][seg][cl] = sm;
  }
  __syncthreads();
  if (seg == 0 && col < H) {
    double tot = 0.0;
    for (int p = 0; p < passes; ++p) {
      double sm = s_s[p][0][cl];
      for (int q = 1; q < FIN_SEGS; ++q) sm += s_s[p][q][cl];
      tot += sm;
    }
    out[col] = (float)tot;
  }
}

static void launch_colsum_final(const float* part, int H, int nc, int passes, float* out, hipStream_t s) {
  const dim3 gr((H + FIN_COLS - 1) / FIN_COLS), bl(TRS_BLOCK);
  const bool two_sweeps = trs_tuning().bn_final_two_sweeps != 0;
  const int per_thread = (nc + FIN_SEGS - 1) / FIN_SEGS;
  if (two_sweeps || passes > 2 || per_thread > 32)
    hipLaunchKernelGGL(colsum_final_kernel, gr, bl, 0, s, part, H, nc, passes, out);
  else if (per_thread <= 16)
    hipLaunchKernelGGL(colsum_final_regs_kernel<16>, gr, bl, 0, s, part, H, nc, passes, out);
  else
    hipLaunchKernelGGL(colsum_final_regs_kernel<32>, gr, bl, 0, s, part, H, nc, passes, out);
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

// ---- float4 variants of the chunk reductions (H % 4 == 0, 16-byte aligned rows).  A workgroup covers TPR*4 columns
// with TPR threads and 256/TPR rows at a time; four row-steps are in flight per thread (unconditional loads), the row
// lanes are merged in LDS in lane order.  Same chunking and order in both passes (exact pos/neg cancellation holds).
struct V4Shape {
  int tpr;  // threads per row (power of two <= 256)
  int gx;   // workgroups across the columns
};
static V4Shape v4_shape(int H) {
  const int c4 = H / 4;
  int tpr = 1;
  while (tpr < c4 && tpr < TRS_BLOCK) tpr <<= 1;
  V4Shape v = {tpr, (c4 + tpr - 1) / tpr};
  return v;
}
static bool v4_ok(const void* p, int H, int64_t ld) { return H % 4 == 0 && ld % 4 == 0 && ((uintptr_t)p & 15) == 0; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// four consecutive elements at offset `off` of an fp32 image or (H16) of a bf16 image of the same shape
template <bool H16>
__device__ __forceinline__ float4 ld4t(const float* p32, const unsigned short* p16, int64_t off) {
  if (H16) {
    const uint2 v = *reinterpret_cast<const uint2*>(p16 + off);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(p32 + off);
}

template <bool Y16, bool DX16, bool OUTER = false>
__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_reduce_v4_kernel(const BnBwdArgs a, int tpr) {
  __shared__ float4 sh[OUTER ? 3 : 2][TRS_BLOCK];
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, nrl = TRS_BLOCK / tpr;
  const int col = (blockIdx.x * tpr + tc) * 4;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  const bool live = col < a.H;
  const int cc = live ? col : 0;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < a.rows_per_pass) ? r0 + CHUNK_ROWS : a.rows_per_pass;
  const int64_t base = (int64_t)pass * a.rows_per_pass;
  const float4 ow4 = OUTER ? *reinterpret_cast<const float4*>(a.ow + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f}, ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.use_bn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mu[q] = a.mean[pass * a.H + cc + q];
      is[q] = 1.0f / sqrtf(a.var[pass * a.H + cc + q] + a.eps);
      ga[q] = a.gamma[cc + q];
      be[q] = a.beta[cc + q];
    }
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, s3[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_xw = OUTER && a.xw_part != nullptr;
  constexpr int U = 4;
  for (int64_t r = r0 + rl; r < r1; r += (int64_t)nrl * U) {
    float4 yv[U], dv[U];
    float grv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      const int64_t rc = rr < r1 ? rr : r1 - 1;
      yv[u] = ld4t<Y16>(a.y, a.y16, (base + rc) * a.ld + cc);
      grv[u] = 0.f;
      if (OUTER) {
        const float gr = a.og[base + rc];
        grv[u] = gr;
        dv[u] = make_float4(gr * ow4.x, gr * ow4.y, gr * ow4.z, gr * ow4.w);
      } else {
        dv[u] = ld4t<DX16>(a.dx, a.dx16, (base + rc) * a.ldd + cc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = r + (int64_t)u * nrl < r1;
      const float y4[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, d4[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xhat = (y4[q] - mu[q]) * is[q];
        const float yhat = a.use_bn ? xhat * ga[q] + be[q] : y4[q];
        const float d = (in && yhat > 0.f) ? d4[q] : 0.f;
        s1[q] += d;
        s2[q] += d * xhat;
        if (OUTER) s3[q] += (in && yhat > 0.f) ? grv[u] * yhat : 0.f;  // og[r] * relu(bn(y)): the forward's activation
      }
    }
  }
  sh[0][threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  sh[1][threadIdx.x] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  if (OUTER) sh[OUTER ? 2 : 0][threadIdx.x] = make_float4(s3[0], s3[1], s3[2], s3[3]);
  __syncthreads();
  if (rl == 0 && live) {
    float4 t1 = sh[0][tc], t2 = sh[1][tc];
    for (int l = 1; l < nrl; ++l) {
      const float4 o1 = sh[0][l * tpr + tc], o2 = sh[1][l * tpr + tc];
      t1.x += o1.x; t1.y += o1.y; t1.z += o1.z; t1.w += o1.w;
      t2.x += o2.x; t2.y += o2.y; t2.z += o2.z; t2.w += o2.w;
    }
    float* o = a.part + (((int64_t)pass * a.n_chunks + chunk) * 2) * a.H;
    *reinterpret_cast<float4*>(o + col) = t1;
    *reinterpret_cast<float4*>(o + a.H + col) = t2;
    if (want_xw) {
      float4 t3 = sh[OUTER ? 2 : 0][tc];
      for (int l = 1; l < nrl; ++l) {
        const float4 o3 = sh[OUTER ? 2 : 0][l * tpr + tc];
        t3.x += o3.x; t3.y += o3.y; t3.z += o3.z; t3.w += o3.w;
      }
      *reinterpret_cast<float4*>(a.xw_part + ((int64_t)pass * a.n_chunks + chunk) * a.H + col) = t3;
    }
  }
}

// out = relu(bn(y)): thread = 4 fixed columns (scale/shift folded once), eight rows in flight per thread, 32 rows per
// workgroup — no reduction here, so the rows are cut fine enough to fill the chip (128-row chunks: 512 workgroups with
// 16 KB in flight per CU ran the 1024-wide bf16 layer at 3.0 TB/s).
constexpr int FWD_ROWS = 32;
template <bool Y16>
__global__ __launch_bounds__(TRS_BLOCK) void bn_relu_fwd_v4_kernel(const BnFwdArgs a, int tpr) {
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, nrl = TRS_BLOCK / tpr;
  const int col = (blockIdx.x * tpr + tc) * 4;
  if (col >= a.H) return;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  const int sp = a.stat_passes > 1 ? pass : 0;
  const int64_t r0 = (int64_t)chunk * FWD_ROWS;
  const int64_t r1 = (r0 + FWD_ROWS < a.rows_per_pass) ? r0 + FWD_ROWS : a.rows_per_pass;
  const int64_t base = (int64_t)pass * a.rows_per_pass;
  float mu[4], is[4], ga[4], be[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    mu[q] = a.use_bn ? a.mean[sp * a.H + col + q] : 0.f;
    is[q] = a.use_bn ? 1.0f / sqrtf(a.var[sp * a.H + col + q] + a.eps) : 1.f;
    ga[q] = a.use_bn ? a.gamma[col + q] : 1.f;
    be[q] = a.use_bn ? a.beta[col + q] : 0.f;
  }
  float dw[4] = {0.f, 0.f, 0.f, 0.f}, db = 0.f;
  if (a.dot_out) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dw[q] = a.dot_w[col + q];
    db = a.dot_b ? a.dot_b[0] : 0.f;
  }
  constexpr int U = 8;
  for (int64_t r = r0 + rl; r < r1; r += (int64_t)nrl * U) {
    float4 yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      yv[u] = ld4t<Y16>(a.y, a.y16, (base + (rr < r1 ? rr : r1 - 1)) * a.ld + col);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      float v[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = a.use_bn ? ((v[q] - mu[q]) * is[q]) * ga[q] + be[q] : v[q];
        v[q] = fmaxf(x, 0.f);
      }
      if (rr < r1) {
        if (a.out) *reinterpret_cast<float4*>(a.out + (base + rr) * a.ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
        if (a.out16) st_bf16x4(a.out16 + (base + rr) * a.ldo + col, v[0], v[1], v[2], v[3]);
      }
      if (a.dot_out) {  // (wave-uniform; the tpr lanes of the row are adjacent lanes of this wave, every one alive)
        float d = ((v[0] * dw[0] + v[1] * dw[1]) + v[2] * dw[2]) + v[3] * dw[3];
        for (int m = tpr >> 1; m > 0; m >>= 1) d += __shfl_xor(d, m);
        if (tc == 0 && rr < r1) a.dot_out[base + rr] = d + db;
      }
    }
  }
  if (a.running_mean && chunk == 0 && pass == 0 && rl == 0) {
    // one thread per 4 columns: the arithmetic of bn_running_update_kernel, pass by pass
    const float n = (float)a.rows_per_pass;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float rm = a.running_mean[col + q], rv = a.running_var[col + q];
      for (int p = 0; p < a.passes; ++p) {
        const float m = a.mean[p * a.H + col + q];
        const double m2 = (double)a.var[p * a.H + col + q] * (double)n;
        const float unb = (float)(m2 / (n > 1.f ? (double)n - 1.0 : 1.0));
        rm = (1.0f - a.momentum) * rm + a.momentum * m;
        rv = (1.0f - a.momentum) * rv + a.momentum * unb;
      }
      a.running_mean[col + q] = rm;
      a.running_var[col + q] = rv;
    }
    if (col == 0 && a.tracked) *a.tracked += a.passes;
  }
}

// dy from (y, dx, final sums) with the same thread-owns-columns layout; optionally the per-chunk column sums of dy.
template <bool Y16, bool DX16, bool OUTER = false>
__global__ __launch_bounds__(TRS_BLOCK) void bn_bwd_apply_v4_kernel(const BnBwdArgs a, int tpr) {
  __shared__ float4 sh[TRS_BLOCK];
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, nrl = TRS_BLOCK / tpr;
  const int col = (blockIdx.x * tpr + tc) * 4;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  const bool live = col < a.H;
  const int cc = live ? col : 0;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < a.rows_per_pass) ? r0 + CHUNK_ROWS : a.rows_per_pass;
  const int64_t base = (int64_t)pass * a.rows_per_pass;
  const float4 ow4 = OUTER ? *reinterpret_cast<const float4*>(a.ow + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float invB = a.inv_n;
  float mu[4], is[4], ga[4], be[4], m1[4], m2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    mu[q] = a.use_bn ? a.mean[pass * a.H + cc + q] : 0.f;
    is[q] = a.use_bn ? 1.0f / sqrtf(a.var[pass * a.H + cc + q] + a.eps) : 1.f;
    ga[q] = a.use_bn ? a.gamma[cc + q] : 1.f;
    be[q] = a.use_bn ? a.beta[cc + q] : 0.f;
    m1[q] = a.use_bn ? a.sums[(pass * 2 + 0) * a.H + cc + q] * invB : 0.f;
    m2[q] = a.use_bn ? a.sums[(pass * 2 + 1) * a.H + cc + q] * invB : 0.f;
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int64_t r = r0 + rl; r < r1; r += (int64_t)nrl * U) {
    float4 yv[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      const int64_t rc = rr < r1 ? rr : r1 - 1;
      yv[u] = ld4t<Y16>(a.y, a.y16, (base + rc) * a.ld + cc);
      if (OUTER) {
        const float gr = a.og[base + rc];
        dv[u] = make_float4(gr * ow4.x, gr * ow4.y, gr * ow4.z, gr * ow4.w);
      } else {
        dv[u] = ld4t<DX16>(a.dx, a.dx16, (base + rc) * a.ldd + cc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      const float y4[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, d4[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (a.use_bn) {
          const float xhat = (y4[q] - mu[q]) * is[q];
          const float yhat = xhat * ga[q] + be[q];
          const float d = yhat > 0.f ? d4[q] : 0.f;
          o[q] = (ga[q] * is[q]) * (d - m1[q] - xhat * m2[q]);
        } else {
          o[q] = y4[q] > 0.f ? d4[q] : 0.f;
        }
      }
      if (rr < r1 && live) {
        if (a.dy) *reinterpret_cast<float4*>(a.dy + (base + rr) * a.ldd + col) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.dy16) st_bf16x4(a.dy16 + (base + rr) * a.ldd + col, o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) cs[q] += o[q];
      }
    }
  }
  if (a.cs_part == nullptr) return;
  sh[threadIdx.x] = make_float4(cs[0], cs[1], cs[2], cs[3]);
  __syncthreads();
  if (rl == 0 && live) {
    float4 t = sh[tc];
    for (int l = 1; l < nrl; ++l) {
      const float4 o = sh[l * tpr + tc];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *reinterpret_cast<float4*>(a.cs_part + ((int64_t)pass * a.n_chunks + chunk) * a.H + col) = t;
  }
}

__global__ __launch_bounds__(TRS_BLOCK) void colsum_partial_v4_kernel(const float* __restrict__ x, int64_t rows_per_pass,
                                                                     int H, int64_t ld, const float* __restrict__ w,
                                                                     int n_chunks, float* __restrict__ part, int tpr) {
  __shared__ float4 sh[TRS_BLOCK];
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, nrl = TRS_BLOCK / tpr;
  const int col = (blockIdx.x * tpr + tc) * 4;
  const int chunk = blockIdx.y, pass = blockIdx.z;
  const bool live = col < H;
  const int cc = live ? col : 0;
  const int64_t base = (int64_t)pass * rows_per_pass;
  const int64_t r0 = (int64_t)chunk * CHUNK_ROWS;
  const int64_t r1 = (r0 + CHUNK_ROWS < rows_per_pass) ? r0 + CHUNK_ROWS : rows_per_pass;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int64_t r = r0 + rl; r < r1; r += (int64_t)nrl * U) {
    float4 xv[U];
    float wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * nrl;
      const int64_t rc = rr < r1 ? rr : r1 - 1;
      xv[u] = ld4(x + (base + rc) * ld + cc);
      wv[u] = w ? w[base + rc] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + (int64_t)u * nrl < r1) {
        if (w) {
          s[0] += wv[u] * xv[u].x; s[1] += wv[u] * xv[u].y; s[2] += wv[u] * xv[u].z; s[3] += wv[u] * xv[u].w;
        } else {
          s[0] += xv[u].x; s[1] += xv[u].y; s[2] += xv[u].z; s[3] += xv[u].w;
        }
      }
    }
  }
  sh[threadIdx.x] = make_float4(s[0], s[1], s[2], s[3]);
  __syncthreads();
  if (rl == 0 && live) {
    float4 t = sh[tc];
    for (int l = 1; l < nrl; ++l) {
      const float4 o = sh[l * tpr + tc];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *reinterpret_cast<float4*>(part + ((int64_t)pass * n_chunks + chunk) * H + col) = t;
  }
}

static int n_chunks_of(int64_t rows) { return (int)((rows + CHUNK_ROWS - 1) / CHUNK_ROWS); }

// ------------------------------------------------------------------------------------------- embedding update
// SGD on the embedding rows of an MLP step straight from d x0 (rows [0,B): positive pass, [B,2B): negative pass; column
// block f = field f).  Two launches instead of one float-atomic scatter per table:
//   user / item   one wave turn per (triple, user | positive item | negative item): the user's two passes are added
//                 in registers (one row update instead of two); rows the duplicate flags call alone in the batch get a
//                 plain read-modify-write, the others float atomics.
//   metadata      few rows, many references each (10 K categories for 65 536 references per column at c5): every
//                 workgroup OWNS a range of categories of every column, scans the batch's metadata ids once (L2), adds
//                 the d x0 segments of the references that name its categories in LDS and applies each owned row once,
//                 atomic-free — the table rows are read and written once per step instead of once per reference.
struct EmbUpdArgs {
  trs_tables T;
  trs_batch Bt;
  const float* dx;            // fp32 d x0 or NULL
  const unsigned short* dx16; // bf16 d x0 or NULL
  int64_t ld;
  float lr;
  const uint8_t* udup;  // per triple: 1 = the user has another reference in the batch (NULL: atomics everywhere)
  const uint8_t* idup;  // per triple x {pos, neg}
  int cats_per_wg;
};

template <bool DX16>
__device__ __forceinline__ float4 ld_dx4(const EmbUpdArgs& a, int64_t off) {
  if (DX16) {
    const uint2 v = *reinterpret_cast<const uint2*>(a.dx16 + off);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xFFFF0000u));
  }
  return *reinterpret_cast<const float4*>(a.dx + off);
}
template <bool DX16>
__device__ __forceinline__ float ld_dx1(const EmbUpdArgs& a, int64_t off) {
  return DX16 ? __uint_as_float((uint32_t)a.dx16[off] << 16) : a.dx[off];
}

// one wave per (triple, user | positive item | negative item) turn: a row of D floats, lane = element (stride 64) —
// float atomics want consecutive lanes on consecutive addresses (one 256-byte request per instruction; four 16-byte-strided
// atomics per lane measured 2.7x slower), and so does the plain read-modify-write of a row that is alone in the batch
template <typename IdT, bool DX16>
__global__ __launch_bounds__(TRS_BLOCK) void mlp_embed_user_item_kernel(const EmbUpdArgs a) {
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.Bt.B;
  const IdT* uid = (const IdT*)a.Bt.user;
  const IdT* pid = (const IdT*)a.Bt.pos;
  const IdT* nid = (const IdT*)a.Bt.neg;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  for (int64_t seg = wave; seg < 3 * B; seg += nwave) {
    const int64_t t = seg / 3;
    const int w = (int)(seg - t * 3);  // 0 user, 1 positive item, 2 negative item
    const IdT* ids = w == 0 ? uid : (w == 1 ? pid : nid);
    const int64_t id = (int64_t)ids[t];
    float* tab = w == 0 ? T.user : T.item;
    const int64_t n_rows = w == 0 ? T.n_users : T.n_items;
    if ((uint64_t)id >= (uint64_t)n_rows) {
      if (lane == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
      continue;
    }
    const bool alone = w == 0 ? (a.udup && a.udup[t] == 0) : (a.idup && a.idup[2 * t + (w - 1)] == 0);
    const int64_t r1 = (w == 2 ? B + t : t) * a.ld + (w == 0 ? 0 : D);  // first (only) d x0 segment
    const int64_t r2 = (B + t) * a.ld;                                   // the user's negative-pass segment
    float* dst = tab + id * (int64_t)D;
    for (int d0 = lane; d0 < D; d0 += 4 * TRS_WAVE) {  // four elements per lane in flight (D = 256: the whole row)
      float g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = d0 + k * TRS_WAVE;
        const int dc = d < D ? d : d0;  // (clamped: no load inside a per-element conditional)
        g[k] = ld_dx1<DX16>(a, r1 + dc);
        const float g2 = ld_dx1<DX16>(a, (w == 0 ? r2 : r1) + dc);
        if (w == 0) g[k] += g2;
      }
      if (alone) {  // (wave-uniform) plain read-modify-write
        float old[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = d0 + k * TRS_WAVE;
          old[k] = dst[d < D ? d : d0];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = d0 + k * TRS_WAVE;
          if (d < D) dst[d] = old[k] - a.lr * g[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = d0 + k * TRS_WAVE;
          if (d < D) atomicAdd(dst + d, -a.lr * g[k]);
        }
      }
    }
  }
}

constexpr int EMB_META_THREADS = 1024, EMB_U = 16, EMB_LIST = 256;
constexpr int EMB_ACC_BYTES = 124 * 1024;  // LDS for the gradient sums (+ 32 KB of per-wave reference lists)
template <typename IdT, bool DX16>
__global__ __launch_bounds__(EMB_META_THREADS) void mlp_embed_meta_kernel(const EmbUpdArgs a) {
  // LDS: [M][C][D] gradient sums | [M][C] touched | per wave: EMB_LIST x {d x0 row, (column << 24) | owned row} | counters.
  // LDS float atomics run at ~2 cycles per LANE on gfx950 (measured here: 214 of 500 us), so the sums are plain
  // read-modify-writes: every owned row belongs to ONE wave (row % 16) and references are routed to their row's wave
  // through per-wave lists (integer LDS atomics, one per matching reference).
  extern __shared__ float emb_acc[];
  const trs_tables& T = a.T;
  const int D = T.D, M = T.M, C = a.cats_per_wg;
  const int B = (int)a.Bt.B;
  constexpr int NW = EMB_META_THREADS / TRS_WAVE;
  int* touched = reinterpret_cast<int*>(emb_acc + (int64_t)M * C * D);
  int* lists = touched + M * C;            // [NW][2][EMB_LIST]
  int* counts = lists + NW * 2 * EMB_LIST;  // [NW] entries appended in this window, [NW] = overflow flag
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < M * C * D; i += EMB_META_THREADS) emb_acc[i] = 0.f;
  for (int i = threadIdx.x; i < M * C; i += EMB_META_THREADS) touched[i] = 0;
  if (threadIdx.x <= NW) counts[threadIdx.x] = 0;
  __syncthreads();
  const int64_t c0 = (int64_t)blockIdx.x * C;  // first owned category (of every column)
  const int chunks = D >> 2;
  const IdT* pm_ids = (const IdT*)a.Bt.pos_meta;
  const IdT* nm_ids = (const IdT*)a.Bt.neg_meta;
  const uint32_t half = (uint32_t)B * (uint32_t)M, total = 2u * half;  // the two (B, M) id matrices, flat
  const uint32_t magic = (uint32_t)(0x100000000ull / (uint32_t)M) + 1u;  // q / M = umulhi(q, magic) for q < 2^32 / M (M > 1)
  uint32_t start = 0, span = EMB_META_THREADS * EMB_U;
  int kept = 0;  // entries of this wave's list from earlier windows (not yet added)
  while (start < total) {  // (workgroup-uniform) windows of flat id positions
    // ---- scan: every workgroup reads ALL metadata ids of the batch (coalesced, EMB_U loads in flight per lane, none
    // inside a conditional block: a block that holds a load ends in s_waitcnt vmcnt(0)) and routes its own
    const uint32_t end = start + span < total ? start + span : total;
    int64_t cat[EMB_U];
#pragma unroll
    for (int k = 0; k < EMB_U; ++k) {
      uint32_t q = start + k * EMB_META_THREADS + threadIdx.x;
      q = q < total ? q : total - 1;
      const IdT* src = q < half ? pm_ids + q : nm_ids + (q - half);
      cat[k] = (int64_t)*src;
    }
#pragma unroll
    for (int k = 0; k < EMB_U; ++k) {
      const uint32_t q = start + k * EMB_META_THREADS + threadIdx.x;
      if (q < end) {
        const uint32_t qq = q < half ? q : q - half;
        const uint32_t r = M == 1 ? qq : __umulhi(qq, magic);
        const int m = (int)(qq - r * (uint32_t)M);
        if ((uint64_t)cat[k] >= (uint64_t)T.n_meta[m]) {
          if (blockIdx.x == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
        } else if (cat[k] >= c0 && cat[k] < c0 + C) {
          const int own = m * C + (int)(cat[k] - c0);
          const int ow = own % NW;
          const int at = atomicAdd(&counts[ow], 1);
          if (at < EMB_LIST) {
            lists[(ow * 2 + 0) * EMB_LIST + at] = (int)((q < half ? 0u : (uint32_t)B) + r);
            lists[(ow * 2 + 1) * EMB_LIST + at] = (m << 24) | own;
          } else {
            counts[NW] = 1;  // a list is full: this window's entries are dropped and it is scanned again (see below)
          }
        }
      }
    }
    __syncthreads();
    const bool overflow = counts[NW] != 0;
    const bool last = !overflow && end >= total;
    // The lists are emptied when one is half full, at the end, and before a window is scanned again — not after every
    // window: a window adds ~4 entries per wave (c5), i.e. one group of row loads whose latency nothing would cover.
    const int mine = overflow ? kept : counts[wave];
    const bool drain = __syncthreads_or((mine > EMB_LIST / 2) || overflow || last) != 0;
    if (!drain) {  // keep collecting
      kept = mine;
      start = end;
      if (span < EMB_META_THREADS * EMB_U) span *= 2;
      continue;
    }
    const int n_list = mine;  // (on overflow: the entries of the earlier windows only)
    // ---- add: each wave sums the d x0 segments of ITS rows' references, eight row loads in flight, plain LDS adds
    const int* list_row = lists + (wave * 2 + 0) * EMB_LIST;
    const int* list_own = lists + (wave * 2 + 1) * EMB_LIST;
    constexpr int AU = 8;
    for (int i = 0; i < n_list; i += AU) {
      float4 g[AU];
      int own[AU];
#pragma unroll
      for (int k = 0; k < AU; ++k) {
        const int e = i + k < n_list ? i + k : n_list - 1;
        const int packed = list_own[e];
        own[k] = packed & 0xFFFFFF;
        const int c = lane < chunks ? lane : 0;
        g[k] = ld_dx4<DX16>(a, (int64_t)list_row[e] * a.ld + (int64_t)(2 + (packed >> 24)) * D + 4 * c);
      }
#pragma unroll
      for (int k = 0; k < AU; ++k) {
        if (i + k < n_list && lane < chunks) {
          float4* acc = reinterpret_cast<float4*>(emb_acc + (int64_t)own[k] * D) + lane;
          float4 v = *acc;
          v.x += g[k].x; v.y += g[k].y; v.z += g[k].z; v.w += g[k].w;
          *acc = v;
        }
      }
      for (int c = lane + TRS_WAVE; c < chunks; c += TRS_WAVE)  // (D > 256: the rest of the row, one entry at a time)
        for (int k = 0; k < AU && i + k < n_list; ++k) {
          const int packed = list_own[i + k];
          const float4 gg = ld_dx4<DX16>(a, (int64_t)list_row[i + k] * a.ld + (int64_t)(2 + (packed >> 24)) * D + 4 * c);
          float4* acc = reinterpret_cast<float4*>(emb_acc + (int64_t)own[k] * D) + c;
          float4 v = *acc;
          v.x += gg.x; v.y += gg.y; v.z += gg.z; v.w += gg.w;
          *acc = v;
        }
    }
    for (int e = lane; e < n_list; e += TRS_WAVE) touched[list_own[e] & 0xFFFFFF] = 1;
    __syncthreads();  // lists and counters are free again
    if (threadIdx.x <= NW) counts[threadIdx.x] = 0;
    kept = 0;
    if (overflow) {
      // the dropped window comes again at half the span, into empty lists (<= EMB_LIST positions always fit)
      span = span > 2 * EMB_LIST ? span / 2 : EMB_LIST;
    } else {
      start = end;
      if (span < EMB_META_THREADS * EMB_U) span *= 2;
    }
    __syncthreads();
  }
  for (int row = wave; row < M * C; row += NW) {  // one owned row per wave turn: a plain read-modify-write
    if (!touched[row]) continue;
    const int m = row / C, lc = row - m * C;
    float* dst = T.meta[m] + (c0 + lc) * (int64_t)D;
    const float* acc = emb_acc + (int64_t)row * D;
    for (int c = lane; c < chunks; c += TRS_WAVE) {
      float4 v = *reinterpret_cast<float4*>(dst + 4 * c);
      const float4 g = *reinterpret_cast<const float4*>(acc + 4 * c);
      v.x -= a.lr * g.x; v.y -= a.lr * g.y; v.z -= a.lr * g.z; v.w -= a.lr * g.w;
      *reinterpret_cast<float4*>(dst + 4 * c) = v;
    }
  }
}

}  // namespace

extern "C" int trs_mlp_gather_concat(const trs_tables* tables, const trs_batch* batch, int32_t passes, float* x_dev,
                                     void* x_bf16_dev, int64_t ld, void* stream) {
  TRS_REQUIRE(tables && batch && (x_dev || x_bf16_dev), "trs_mlp_gather_concat: NULL argument");
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
  GatherArgs a = {*tables, *batch, x_dev, ld, passes, (unsigned short*)x_bf16_dev};
  const int64_t segs = (int64_t)passes * batch->B * (2 + tables->M);
  int g_lanes = 1;
  while (g_lanes < 64 && g_lanes < (tables->D + 3) / 4) g_lanes <<= 1;
  hipLaunchKernelGGL(mlp_gather_kernel, dim3(trs_grid((segs + 3) / 4, TRS_BLOCK / g_lanes)), dim3(TRS_BLOCK), 0,
                     (hipStream_t)stream, a);
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
  launch_bn_stats_final(workspace_dev, rows_per_pass, CHUNK_ROWS, H, nc, passes, mean_out_dev, var_out_dev, s);
  TRS_CHECK_LAUNCH("bn_stats_final_kernel");
  if (running_mean_dev) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((H + TRS_BLOCK - 1) / TRS_BLOCK), dim3(TRS_BLOCK), 0, s,
                       mean_out_dev, var_out_dev, rows_per_pass, H, passes, momentum, running_mean_dev, running_var_dev,
                       (long long*)nullptr);
    TRS_CHECK_LAUNCH("bn_running_update_kernel");
  }
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
  launch_bn_stats_final(part_dev, rows_per_pass, chunk_rows, H, nc, passes, mean_out_dev, var_out_dev, (hipStream_t)stream);
  TRS_CHECK_LAUNCH("bn_stats_final_kernel");
  if (running_mean_dev) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((H + TRS_BLOCK - 1) / TRS_BLOCK), dim3(TRS_BLOCK), 0,
                       (hipStream_t)stream, mean_out_dev, var_out_dev, rows_per_pass, H, passes, momentum,
                       running_mean_dev, running_var_dev, (long long*)nullptr);
    TRS_CHECK_LAUNCH("bn_running_update_kernel");
  }
  return TRS_OK;
}

extern "C" int trs_bn_relu_forward(const void* y_dev, int32_t y_bf16, int64_t rows_per_pass, int32_t passes, int32_t H,
                                   int64_t ld, int32_t use_bn, int32_t stat_passes, const float* mean_dev,
                                   const float* var_dev, const float* gamma_dev, const float* beta_dev, float eps,
                                   float* out_dev, void* out_bf16_dev, int64_t ldo, float momentum,
                                   float* running_mean_dev, float* running_var_dev, int64_t* num_batches_tracked_dev,
                                   const float* dot_w_dev, const float* dot_bias_dev, float* dot_out_dev, void* stream) {
  TRS_REQUIRE(!num_batches_tracked_dev || running_mean_dev, "trs_bn_relu_forward: the batch counter rides with the running update");
  TRS_REQUIRE(y_dev && (out_dev || out_bf16_dev || dot_out_dev), "trs_bn_relu_forward: NULL argument");
  TRS_REQUIRE((dot_w_dev == nullptr) == (dot_out_dev == nullptr),
              "trs_bn_relu_forward: the output-layer dot needs its weights and its output");
  TRS_REQUIRE((running_mean_dev == nullptr) == (running_var_dev == nullptr), "trs_bn_relu_forward: running stats");
  TRS_REQUIRE(!running_mean_dev || (use_bn && stat_passes == passes),
              "trs_bn_relu_forward: the running update needs the batch statistics of every pass");
  TRS_REQUIRE(rows_per_pass >= 0 && H > 0 && ld >= H && ldo >= H && passes >= 1, "trs_bn_relu_forward: bad shape");
  TRS_REQUIRE(!use_bn || (mean_dev && var_dev && gamma_dev && beta_dev), "trs_bn_relu_forward: BN tensors are NULL");
  TRS_REQUIRE(stat_passes == 1 || stat_passes == passes, "trs_bn_relu_forward: stat_passes must be 1 or passes");
  if (rows_per_pass == 0) return TRS_OK;
  BnFwdArgs a = {y_bf16 ? nullptr : (const float*)y_dev, out_dev, (unsigned short*)out_bf16_dev, rows_per_pass, ld,
                 ldo, H, passes, stat_passes, use_bn, mean_dev, var_dev, gamma_dev, beta_dev, eps,
                 y_bf16 ? (const unsigned short*)y_dev : nullptr, momentum, running_mean_dev, running_var_dev,
                 (long long*)num_batches_tracked_dev, nullptr, nullptr, nullptr};
  const bool v4 = H % 4 == 0 && ld % 4 == 0 && ((uintptr_t)y_dev & (y_bf16 ? 7 : 15)) == 0 &&
                  (!out_dev || v4_ok(out_dev, H, ldo)) &&
                  (!out_bf16_dev || (ldo % 4 == 0 && ((uintptr_t)out_bf16_dev & 7) == 0));
  TRS_REQUIRE(v4 || (out_dev && !out_bf16_dev && !y_bf16),
              "trs_bn_relu_forward: bf16 images need H %% 4 == 0 and aligned rows");
  const V4Shape v = v4_shape(H % 4 == 0 ? H : 4);
  // the dot inside the launch: a row = adjacent lanes of one wave
  const bool dot_fused = dot_out_dev && v4 && v.gx == 1 && v.tpr <= TRS_WAVE && 4 * v.tpr == H;
  TRS_REQUIRE(out_dev || out_bf16_dev || dot_fused,
              "trs_bn_relu_forward: the dot of this layer (H = %d) is not formed inside the launch: it needs out_dev", H);
  TRS_REQUIRE(!dot_out_dev || dot_fused || out_dev, "trs_bn_relu_forward: the row-dot kernel reads the fp32 out_dev");
  if (v4) {
    if (dot_fused) {
      a.dot_w = dot_w_dev; a.dot_b = dot_bias_dev; a.dot_out = dot_out_dev;
    }
    const dim3 gr(v.gx, (unsigned)((rows_per_pass + FWD_ROWS - 1) / FWD_ROWS), passes);
    if (y_bf16) hipLaunchKernelGGL(bn_relu_fwd_v4_kernel<true>, gr, dim3(TRS_BLOCK), 0, (hipStream_t)stream, a, v.tpr);
    else hipLaunchKernelGGL(bn_relu_fwd_v4_kernel<false>, gr, dim3(TRS_BLOCK), 0, (hipStream_t)stream, a, v.tpr);
  } else {
    const int64_t total = rows_per_pass * passes * ((H + 3) / 4);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(trs_grid(total, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream, a);
    if (running_mean_dev)
      hipLaunchKernelGGL(bn_running_update_kernel, dim3((H + TRS_BLOCK - 1) / TRS_BLOCK), dim3(TRS_BLOCK), 0,
                         (hipStream_t)stream, mean_dev, var_dev, rows_per_pass, H, passes, momentum, running_mean_dev,
                         running_var_dev, (long long*)num_batches_tracked_dev);
  }
  TRS_CHECK_LAUNCH("bn_relu_fwd_kernel");
  if (dot_out_dev && !dot_fused)  // wide or unaligned layers: the stand-alone row dot on the fp32 output
    return trs_rowdot(out_dev, rows_per_pass * passes, H, ldo, dot_w_dev, dot_bias_dev, dot_out_dev, stream);
  return TRS_OK;
}

extern "C" int trs_bn_relu_backward(const void* y_dev, int32_t y_bf16, const void* dx_dev, int32_t dx_bf16,
                                    int64_t rows_per_pass, int32_t passes,
                                    int32_t H, int64_t ld, int64_t ldd, int32_t use_bn, const float* mean_dev,
                                    const float* var_dev, const float* gamma_dev, const float* beta_dev, float eps,
                                    float* dy_dev, void* dy_bf16_dev, float* dgamma_dev, float* dbeta_dev,
                                    float* dy_colsum_dev, float* workspace_dev, int32_t phase, float* sums_dev,
                                    int64_t stat_rows, const float* outer_g_dev, const float* outer_w_dev,
                                    float* outer_xw_dev, void* stream) {
  const bool outer = outer_g_dev != nullptr;
  TRS_REQUIRE(!outer_xw_dev || (outer && use_bn && phase == 0),
              "trs_bn_relu_backward: outer_xw comes out of the outer-product form's BatchNorm reduce (phase 0)");
  TRS_REQUIRE((outer_g_dev == nullptr) == (outer_w_dev == nullptr) && (!outer || (!dx_dev && !dx_bf16)),
              "trs_bn_relu_backward: the outer-product form takes (outer_g, outer_w) INSTEAD of dx");
  TRS_REQUIRE(y_dev && (dx_dev || outer) && (dy_dev || dy_bf16_dev || phase == 1) && workspace_dev,
              "trs_bn_relu_backward: NULL argument");
  TRS_REQUIRE(phase >= 0 && phase <= 2 && (phase == 0 || (sums_dev && use_bn)) && stat_rows >= 0,
              "trs_bn_relu_backward: phases 1 / 2 need BatchNorm and the (passes,2,H) sums buffer");
  TRS_REQUIRE(rows_per_pass > 0 && H > 0 && ld >= H && ldd >= H && passes >= 1 && passes <= 2,
              "trs_bn_relu_backward: bad shape");
  TRS_REQUIRE(!use_bn || (mean_dev && var_dev && gamma_dev && beta_dev), "trs_bn_relu_backward: BN tensors are NULL");
  const int nc = n_chunks_of(rows_per_pass);
  const int gx = (H + TRS_BLOCK - 1) / TRS_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  float* sums = sums_dev ? sums_dev : workspace_dev + (int64_t)passes * nc * 2 * H;  // (passes,2,H) behind the partials
  float* cs_part = workspace_dev + (int64_t)passes * nc * 2 * H + (int64_t)passes * 2 * H;  // (passes,nc,H) column-sum partials of dy
  float* xw_part = cs_part + (int64_t)passes * nc * H;  // (passes,nc,H) partials of outer_xw
  const bool h16 = y_bf16 || dx_bf16;
  BnBwdArgs a = {y_bf16 ? nullptr : (const float*)y_dev, dx_bf16 ? nullptr : (const float*)dx_dev, dy_dev,
                 rows_per_pass, ld, ldd, H, passes, use_bn, nc, mean_dev, var_dev, gamma_dev, beta_dev, eps,
                 workspace_dev, sums, dy_colsum_dev ? cs_part : nullptr, (unsigned short*)dy_bf16_dev,
                 y_bf16 ? (const unsigned short*)y_dev : nullptr, dx_bf16 ? (const unsigned short*)dx_dev : nullptr,
                 1.0f / (float)(stat_rows > 0 ? stat_rows : rows_per_pass), outer_g_dev, outer_w_dev,
                 outer_xw_dev ? xw_part : nullptr};
  const bool v4in = H % 4 == 0 && ld % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)y_dev & (y_bf16 ? 7 : 15)) == 0 &&
                    ((uintptr_t)dx_dev & (dx_bf16 ? 7 : 15)) == 0 && v4_ok(workspace_dev, H, 4) &&
                    (!outer || ((uintptr_t)outer_w_dev & 15) == 0);
  TRS_REQUIRE(!outer || (v4in && use_bn >= 0), "trs_bn_relu_backward: the outer-product form needs H %% 4 == 0 and aligned rows");
  const bool v4a = v4in && (!dy_dev || v4_ok(dy_dev, H, ldd)) && (!dy_bf16_dev || ((uintptr_t)dy_bf16_dev & 7) == 0);
  TRS_REQUIRE(v4a || phase == 1 || (dy_dev && !dy_bf16_dev && !h16),
              "trs_bn_relu_backward: bf16 images need H %% 4 == 0 and aligned rows");
  TRS_REQUIRE(phase != 1 || v4in || !h16, "trs_bn_relu_backward: bf16 images need H %% 4 == 0 and aligned rows");
  const V4Shape v = v4_shape(H % 4 == 0 ? H : 4);
  const dim3 g4(v.gx, nc, passes), bl(TRS_BLOCK);
  // (y, dx) image types of the bf16-resident path: (bf16, bf16) inside the net, (bf16, fp32) at the last hidden layer
  // (its dx comes from the fp32 H -> 1 output layer)
#define TRS_BWD(KERNEL)                                                                             \
  {                                                                                                 \
    if (outer && y_bf16) hipLaunchKernelGGL((KERNEL<true, false, true>), g4, bl, 0, s, a, v.tpr);   \
    else if (outer) hipLaunchKernelGGL((KERNEL<false, false, true>), g4, bl, 0, s, a, v.tpr);       \
    else if (y_bf16 && dx_bf16) hipLaunchKernelGGL((KERNEL<true, true>), g4, bl, 0, s, a, v.tpr);   \
    else if (y_bf16) hipLaunchKernelGGL((KERNEL<true, false>), g4, bl, 0, s, a, v.tpr);             \
    else hipLaunchKernelGGL((KERNEL<false, false>), g4, bl, 0, s, a, v.tpr);                        \
  }
  TRS_REQUIRE(!dx_bf16 || y_bf16, "trs_bn_relu_backward: a bf16 dx comes with a bf16 y");
  if (use_bn && phase != 2) {
    if (v4in) {
      TRS_BWD(bn_bwd_reduce_v4_kernel)
    } else {
      hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, a);
    }
    TRS_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    launch_bn_bwd_final(workspace_dev, H, nc, passes, sums, dgamma_dev, dbeta_dev, s);
    TRS_CHECK_LAUNCH("bn_bwd_final_kernel");
    if (outer_xw_dev) {
      launch_colsum_final(xw_part, H, nc, passes, outer_xw_dev, s);
      TRS_CHECK_LAUNCH("colsum_final_kernel");
    }
  }
  if (phase == 1) return TRS_OK;  // the caller reduces `sums` over ranks, then calls phase 2 with stat_rows = world * rows
  if (v4a) {
    TRS_BWD(bn_bwd_apply_v4_kernel)
    TRS_CHECK_LAUNCH("bn_bwd_apply_kernel");
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(trs_grid(rows_per_pass * passes * H, TRS_BLOCK)), dim3(TRS_BLOCK), 0,
                       s, a);
    TRS_CHECK_LAUNCH("bn_bwd_apply_kernel");
    if (dy_colsum_dev) {
      hipLaunchKernelGGL(colsum_partial_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, dy_dev, rows_per_pass, H,
                         ldd, (const float*)nullptr, nc, cs_part);
      TRS_CHECK_LAUNCH("colsum_partial_kernel");
    }
  }
#undef TRS_BWD
  if (dy_colsum_dev) {
    launch_colsum_final(cs_part, H, nc, passes, dy_colsum_dev, s);
    TRS_CHECK_LAUNCH("colsum_final_kernel");
  }
  return TRS_OK;
}

extern "C" int64_t trs_bn_backward_workspace_floats(int64_t rows_per_pass, int32_t H, int32_t passes) {
  return (int64_t)passes * n_chunks_of(rows_per_pass) * 4 * H + (int64_t)passes * 2 * H;
}

extern "C" int trs_colsum(const float* x_dev, int64_t rows_per_pass, int32_t passes, int32_t H, int64_t ld,
                          const float* row_weight_dev, float* out_dev, float* workspace_dev, void* stream) {
  TRS_REQUIRE(x_dev && out_dev && workspace_dev, "trs_colsum: NULL argument");
  TRS_REQUIRE(rows_per_pass > 0 && passes >= 1 && H > 0 && ld >= H, "trs_colsum: bad shape");
  const int nc = n_chunks_of(rows_per_pass);
  const int gx = (H + TRS_BLOCK - 1) / TRS_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  if (v4_ok(x_dev, H, ld) && v4_ok(workspace_dev, H, 4)) {
    const V4Shape v = v4_shape(H);
    hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(v.gx, nc, passes), dim3(TRS_BLOCK), 0, s, x_dev, rows_per_pass, H,
                       ld, row_weight_dev, nc, workspace_dev, v.tpr);
  } else {
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(gx, nc, passes), dim3(TRS_BLOCK), 0, s, x_dev, rows_per_pass, H, ld,
                       row_weight_dev, nc, workspace_dev);
  }
  TRS_CHECK_LAUNCH("colsum_partial_kernel");
  launch_colsum_final(workspace_dev, H, nc, passes, out_dev, s);
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

// owned categories per workgroup of mlp_embed_meta_kernel: as many as 128 KB of LDS accumulators hold, but no fewer
// workgroups than CUs; every workgroup scans all metadata ids of the batch, so the form stops paying beyond ~1000 of them
static bool embed_meta_plan(const trs_tables* tables, int64_t* cats_per_wg, int64_t* wgs) {
  const int D = tables->D, M = tables->M;
  int64_t max_cat = 0;
  for (int m = 0; m < M; ++m) max_cat = tables->n_meta[m] > max_cat ? tables->n_meta[m] : max_cat;
  if (M == 0 || max_cat <= 0) return false;
  const int64_t c_lds = EMB_ACC_BYTES / ((int64_t)M * (D * 4 + 4));
  int64_t c = (max_cat + 255) / 256;
  if (c > c_lds) c = c_lds;
  if (c < 1) return false;
  *cats_per_wg = c;
  *wgs = (max_cat + c - 1) / c;
  return *wgs <= 1024;
}

extern "C" int trs_mlp_embed_sgd_update_supported(const trs_tables* tables) {
  if (!tables || tables->D <= 0 || tables->D % 4 != 0 || tables->M < 0 || tables->M > TRS_MAX_META) return 0;
  int64_t c, w;
  return tables->M == 0 || embed_meta_plan(tables, &c, &w) ? 1 : 0;
}

// SGD update of every embedding table of an MLP step from d x0 (fp32 or bf16; (2B, ld), column block f = field f):
// see mlp_embed_user_item_kernel / mlp_embed_meta_kernel.  user_dup_flags (B) / item_dup_flags (B, 2) are optional
// (trs_epoch_flags / trs_epoch_presort produce them): NULL = float atomics for every user / item reference.
extern "C" int trs_mlp_embed_sgd_update(const trs_tables* tables, const trs_batch* batch, const float* dx0_dev,
                                        const void* dx0_bf16_dev, int64_t ld, float lr,
                                        const uint8_t* user_dup_flags_dev, const uint8_t* item_dup_flags_dev,
                                        void* stream) {
  TRS_REQUIRE(tables && batch && ((dx0_dev != nullptr) != (dx0_bf16_dev != nullptr)),
              "trs_mlp_embed_sgd_update: NULL argument (exactly one of dx0_dev / dx0_bf16_dev)");
  TRS_REQUIRE(tables->M >= 0 && tables->M <= TRS_MAX_META && tables->user && tables->item,
              "trs_mlp_embed_sgd_update: bad tables");
  TRS_REQUIRE(tables->D > 0 && tables->D % 4 == 0 && ld % 4 == 0 && ld >= (int64_t)(2 + tables->M) * tables->D,
              "trs_mlp_embed_sgd_update: needs D and ld multiples of 4, ld >= (2 + M) * D");
  TRS_REQUIRE(((uintptr_t)(dx0_dev ? (const void*)dx0_dev : dx0_bf16_dev) & 15) == 0,
              "trs_mlp_embed_sgd_update: d x0 must be 16-byte aligned");
  TRS_REQUIRE(batch->idx_bytes == 4 || batch->idx_bytes == 8, "trs_mlp_embed_sgd_update: idx_bytes must be 4 or 8");
  if (batch->B == 0) return TRS_OK;
  TRS_REQUIRE(batch->B < ((int64_t)1 << 30) && 2 * batch->B * tables->M * tables->M < ((int64_t)1 << 32),
              "trs_mlp_embed_sgd_update: batch too large");
  TRS_REQUIRE(batch->user && batch->pos && batch->neg, "trs_mlp_embed_sgd_update: ids are NULL");
  TRS_REQUIRE(tables->M == 0 || (batch->pos_meta && batch->neg_meta), "trs_mlp_embed_sgd_update: metadata ids NULL");
  for (int m = 0; m < tables->M; ++m) TRS_REQUIRE(tables->meta[m], "trs_mlp_embed_sgd_update: metadata table %d NULL", m);
  EmbUpdArgs a = {};
  a.T = *tables;
  a.Bt = *batch;
  a.dx = dx0_dev;
  a.dx16 = (const unsigned short*)dx0_bf16_dev;
  a.ld = ld;
  a.lr = lr;
  a.udup = user_dup_flags_dev;
  a.idup = item_dup_flags_dev;
  hipStream_t s = (hipStream_t)stream;
  const int D = tables->D, M = tables->M;
  const dim3 grid_ui(trs_grid(batch->B * 3, TRS_BLOCK / TRS_WAVE));
  const bool i64 = batch->idx_bytes == 8, b16 = dx0_bf16_dev != nullptr;
#define TRS_EMB(KERNEL, GRID, BLOCK, LDS)                                                            \
  {                                                                                                  \
    if (i64 && b16) hipLaunchKernelGGL((KERNEL<int64_t, true>), GRID, BLOCK, LDS, s, a);             \
    else if (i64) hipLaunchKernelGGL((KERNEL<int64_t, false>), GRID, BLOCK, LDS, s, a);              \
    else if (b16) hipLaunchKernelGGL((KERNEL<int32_t, true>), GRID, BLOCK, LDS, s, a);               \
    else hipLaunchKernelGGL((KERNEL<int32_t, false>), GRID, BLOCK, LDS, s, a);                       \
  }
  TRS_EMB(mlp_embed_user_item_kernel, grid_ui, dim3(TRS_BLOCK), 0)
  TRS_CHECK_LAUNCH("mlp_embed_user_item_kernel");
  if (M == 0) return TRS_OK;
  int64_t c, wgs;
  TRS_REQUIRE(embed_meta_plan(tables, &c, &wgs),
              "trs_mlp_embed_sgd_update: metadata tables too large for the owner-computes update (ask "
              "trs_mlp_embed_sgd_update_supported first and scatter per table with trs_rows_scatter_add)");
  a.cats_per_wg = (int)c;
  const size_t lds = (size_t)M * c * (D * 4 + 4) + (size_t)(EMB_META_THREADS / TRS_WAVE) * 2 * EMB_LIST * 4 +
                     (size_t)(EMB_META_THREADS / TRS_WAVE + 1) * 4;
  static const int attr_done = [] {  // (a function-local static: initialised once, thread-safe)
    const int cap = 160 * 1024 - 1024;
    (void)hipFuncSetAttribute((const void*)mlp_embed_meta_kernel<int64_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)mlp_embed_meta_kernel<int64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)mlp_embed_meta_kernel<int32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)mlp_embed_meta_kernel<int32_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    return 1;
  }();
  (void)attr_done;
  TRS_EMB(mlp_embed_meta_kernel, dim3((unsigned)wgs), dim3(EMB_META_THREADS), lds)
#undef TRS_EMB
  TRS_CHECK_LAUNCH("mlp_embed_meta_kernel");
  return TRS_OK;
}
