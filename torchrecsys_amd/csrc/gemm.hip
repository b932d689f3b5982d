// gemm.hip — fp32-in / fp32-accumulate GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: an exact fp32 FMA
// chain, bit-compatible with a k-ordered fmaf loop) for the MLP scorer's dense layers (reference
// collaborative/mlp.py:107-113: nn.Linear = aten::addmm, and its autograd mm's).
//
// C(M,N) = alpha * op(A)(M,K) * op(B)(K,N) + beta * C  [+ bias(N)]          row-major storage
//   transA = 0: A stored (M,K)   (k contiguous)      transA = 1: A stored (K,M)   (m contiguous)
//   transB = 0: B stored (K,N)   (n contiguous)      transB = 1: B stored (N,K)   (k contiguous)
//   forward  y  = x  W^T : transA 0, transB 1        dgrad dx = dy W : transA 0, transB 0
//   wgrad    dW = dy^T x : transA 1, transB 0  (K = rows of the batch: split-K over workgroups, slabs + reduce)
//
// Tiling: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of 32x32, 64 accumulator
// VGPRs); operand tiles staged global -> registers -> LDS as [k][m|n] images (conflict-free ds_read_b32 for the
// 32-lane MFMA fragments), double-buffered so tile t+1's global loads fly under tile t's MFMAs.
#include "trs_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = 132;  // max of the two image strides below
// image stride per staging form: k-contiguous sources are transposed on the LDS write with 4 ds_write_b32 per
// float4 — stride 129 (odd) spreads the 8 k-chunks x 4 rows of a 32-lane group over 32 banks; m/n-contiguous sources
// go in with one aligned ds_write_b128 — stride 132.
__host__ __device__ constexpr int img_ld(bool kcontig) { return kcontig ? 129 : 132; }

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // (N) added to every row, or NULL
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  float alpha, beta;
  int64_t k_per_split;  // K range per split (multiple of BK)
  float* slabs;         // (splits, M, N) partial products when splits > 1
  int vecA, vecB;       // 16-byte global loads legal (alignment + leading dimension)
  int gx, gy, splits;   // logical grid: gx column tiles x gy row tiles x splits, launched as one 1-D grid
  // fused BatchNorm batch statistics of the output (forward GEMMs of the MLP): per 128-row tile and column the mean
  // and the sum of squared deviations from that mean, laid out (row tile, 2, N) — the chunk partials that
  // trs_bn_stats_finalize combines (Chan).  NULL = off.  Requires splits == 1.
  float* bn_part;
  unsigned short* C16;  // bf16 output instead of C (bf16-resident path, no split-K): row stride ldc elements
};

// Global -> registers for one 128 x 32 operand tile.  KC: source is k-contiguous.
//   KC:  chunk c in [0,1024): row = c / 8, k4 = c % 8       (8 float4 per row of 32 k)
//   !KC: chunk c in [0,1024): k = c / 32, r4 = c % 32       (32 float4 per k-row of 128 m/n)
template <bool KC>
__device__ __forceinline__ void tile_load(float4 (&r)[4], const float* __restrict__ P, int64_t ld, int64_t rows,
                                          int64_t Kend, int64_t row0, int64_t k0, int vec, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + 256 * it;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int64_t row = row0 + (c >> 3), k = k0 + ((c & 7) << 2);
      if (row < rows && k < Kend) {
        const float* p = P + row * ld + k;
        if (vec && k + 3 < Kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (k + 1 < Kend) v.y = p[1];
          if (k + 2 < Kend) v.z = p[2];
          if (k + 3 < Kend) v.w = p[3];
        }
      }
    } else {
      const int64_t k = k0 + (c >> 5), row = row0 + ((c & 31) << 2);
      if (k < Kend && row < rows) {
        const float* p = P + k * ld + row;
        if (vec && row + 3 < rows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (row + 1 < rows) v.y = p[1];
          if (row + 2 < rows) v.z = p[2];
          if (row + 3 < rows) v.w = p[3];
        }
      }
    }
    r[it] = v;
  }
}

// Interior tiles (whole 128 x 32 tile in range, 16-byte loads legal): four unconditional float4 loads, nothing under a
// branch, so the compiler can leave them in flight across the MFMA block (a load inside a conditional is followed by
// s_waitcnt vmcnt(0) at the join).
template <bool KC>
__device__ __forceinline__ void tile_load_fast(float4 (&r)[4], const float* __restrict__ P, int64_t ld, int64_t row0,
                                               int64_t k0, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + 256 * it;
    const float* p = KC ? P + (row0 + (c >> 3)) * ld + k0 + ((c & 7) << 2)
                        : P + (k0 + (c >> 5)) * ld + row0 + ((c & 31) << 2);
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    r[it] = make_float4(v.x, v.y, v.z, v.w);
  }
}

template <bool KC>
__device__ __forceinline__ void tile_store(const float4 (&r)[4], float* __restrict__ S, int tid) {
  constexpr int LD = img_ld(KC);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + 256 * it;
    if (KC) {
      const int row = c >> 3, k = (c & 7) << 2;
      S[(k + 0) * LD + row] = r[it].x;
      S[(k + 1) * LD + row] = r[it].y;
      S[(k + 2) * LD + row] = r[it].z;
      S[(k + 3) * LD + row] = r[it].w;
    } else {
      const int k = c >> 5, row = (c & 31) << 2;
      const f32x4 v = {r[it].x, r[it].y, r[it].z, r[it].w};  // native vector: one ds_write_b128 straight from VGPRs
      *reinterpret_cast<f32x4*>(&S[k * LD + row]) = v;
    }
  }
}

// epilogue shared by the fp32 and bf16 kernels: C/D map of the 32x32 MFMA (dtype-independent on gfx950):
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <bool OUT16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2], int64_t m0, int64_t n0,
                                              int wm, int wn, int lr, int lk, int bz) {
  const bool split = g.splits > 1;
  float* out = split ? g.slabs + (int64_t)bz * g.M * g.N : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + lr;
      if (col >= g.N) continue;
      const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        if (!split) {
          v = g.alpha * v + bv;
          if (g.beta != 0.f) v += g.beta * out[row * ldo + col];
        }
        if (OUT16) g.C16[row * ldo + col] = __builtin_bit_cast(unsigned short, (__bf16)v);  // never with split-K
        else out[row * ldo + col] = v;
      }
    }
}

// Per-tile column statistics of the values the epilogue stores (alpha*acc + bias): each column of the 128-row tile is
// spread over 2 waves (wm) x 2 lane halves x 32 accumulator registers; sums meet through a lane-half shuffle and LDS.
// A 256-row workgroup tile (bf16-resident kernel, 8 waves) is two independent 128-row halves: half = wm >> 1 has its own
// LDS cells and its own partial (the chunk size of the statistics stays 128 rows).
__device__ __forceinline__ void gemm_tile_bn_stats(const GemmArgs& g, const f32x16 (&acc)[2][2], float* __restrict__ lds,
                                                   int64_t m0, int64_t n0, int wm, int wn, int lr, int lk, int by,
                                                   int halves = 1, int wn_count = 2) {
  const int half = wm >> 1, wmh = wm & 1;
  m0 += half * BM;
  by = by * halves + half;
  lds += half * wn_count * 128;  // wn_count (wn) x 2 (j) x 32 (lr) x 2 (wmh) cells per half
  wm = wmh;
  const int64_t rows_left = g.M - m0;
  const float n_rows = (float)(rows_left < BM ? rows_left : BM);
  float mean[2];
  __syncthreads();  // the operand images are dead: reuse their LDS
#pragma unroll
  for (int phase = 0; phase < 2; ++phase) {  // 0: sums -> means, 1: squared deviations
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + lr;
      const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (row < g.M) {
            const float v = g.alpha * acc[i][j][r] + bv;
            if (phase == 0) s += v; else { const float d = v - mean[j]; s += d * d; }
          }
        }
      s += __shfl_xor(s, 32, 64);
      if (lk == 0) lds[((wn * 2 + j) * 32 + lr) * 2 + wm] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float tot = lds[((wn * 2 + j) * 32 + lr) * 2 + 0] + lds[((wn * 2 + j) * 32 + lr) * 2 + 1];
      if (phase == 0) {
        mean[j] = tot / n_rows;
      } else if (wm == 0 && lk == 0) {
        const int64_t col = n0 + wn * 64 + j * 32 + lr;
        if (col < g.N) {
          float* o = g.bn_part + (int64_t)by * 2 * g.N;
          o[col] = mean[j];
          o[g.N + col] = tot;
        }
      }
    }
    __syncthreads();
  }
}

template <bool AKC, bool BKC, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDS_LD];
  constexpr int LDA_S = img_ld(AKC), LDB_S = img_ld(BKC);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so dispatch id b
  // is remapped to logical id (b % 8) * (n / 8) + b / 8: every XCD walks a contiguous range of logical tiles, and the
  // gx column tiles that share one A row panel are fetched through one L2 instead of up to gx of them.
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int bx = (int)(lid % g.gx);
  const int by = (int)((lid / g.gx) % g.gy);
  const int bz = (int)(lid / ((int64_t)g.gx * g.gy));
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
  const int64_t kbeg = (int64_t)bz * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  const int64_t nk = (kend - kbeg + BK - 1) / BK;
  const int lr = lane & 31, lk = lane >> 5;
  if (FAST) {
    // every tile interior (host-checked): tile t+1's loads are issued unconditionally before tile t's MFMAs (the last
    // iteration re-reads its own tile into the idle buffer instead of branching around the loads)
    tile_load_fast<AKC>(ra, g.A, g.lda, m0, kbeg, tid);
    tile_load_fast<BKC>(rb, g.B, g.ldb, n0, kbeg, tid);
    tile_store<AKC>(ra, As[0], tid);
    tile_store<BKC>(rb, Bs[0], tid);
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
      const int cur = (int)(kt & 1);
      const int64_t kn = kbeg + (kt + 1 < nk ? kt + 1 : kt) * BK;
      tile_load_fast<AKC>(ra, g.A, g.lda, m0, kn, tid);
      tile_load_fast<BKC>(rb, g.B, g.ldb, n0, kn, tid);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMA block (the scheduler sinks them to their use)
      const float* as = As[cur] + wm * 64 + lr;
      const float* bs = Bs[cur] + wn * 64 + lr;
#pragma unroll
      for (int k2 = 0; k2 < BK / 2; ++k2) {
        const int k = 2 * k2 + lk;
        const float a0 = as[k * LDA_S], a1 = as[k * LDA_S + 32];
        const float b0 = bs[k * LDB_S], b1 = bs[k * LDB_S + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      tile_store<AKC>(ra, As[cur ^ 1], tid);
      tile_store<BKC>(rb, Bs[cur ^ 1], tid);
      __syncthreads();
    }
  } else {
  if (nk > 0) {
    tile_load<AKC>(ra, g.A, g.lda, g.M, kend, m0, kbeg, g.vecA, tid);
    tile_load<BKC>(rb, g.B, g.ldb, g.N, kend, n0, kbeg, g.vecB, tid);
    tile_store<AKC>(ra, As[0], tid);
    tile_store<BKC>(rb, Bs[0], tid);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) {
      tile_load<AKC>(ra, g.A, g.lda, g.M, kend, m0, kbeg + (kt + 1) * BK, g.vecA, tid);
      tile_load<BKC>(rb, g.B, g.ldb, g.N, kend, n0, kbeg + (kt + 1) * BK, g.vecB, tid);
    }
    const float* as = As[cur] + wm * 64 + lr;
    const float* bs = Bs[cur] + wn * 64 + lr;
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const int k = 2 * k2 + lk;
      const float a0 = as[k * LDA_S], a1 = as[k * LDA_S + 32];
      const float b0 = bs[k * LDB_S], b1 = bs[k * LDB_S + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      tile_store<AKC>(ra, As[cur ^ 1], tid);
      tile_store<BKC>(rb, Bs[cur ^ 1], tid);
    }
    __syncthreads();
  }
  }

  gemm_epilogue(g, acc, m0, n0, wm, wn, lr, lk, bz);
  if (g.bn_part) gemm_tile_bn_stats(g, acc, &As[0][0], m0, n0, wm, wn, lr, lk, by);
}

// ---- bf16-input variant (use_amp): operands are read as fp32, rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they
// are staged into LDS, multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) and accumulated in fp32.
// LDS images are [m|n][k] with k contiguous (a lane's fragment = 8 consecutive k = one ds_read_b128), rows padded to
// 40 bf16 = 80 B so the 16 lanes of a ds_read_b128 group land on 16 distinct 16-B slots.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int HROW = 40;  // bf16 elements per LDS row (BK = 32 + 8 pad)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

template <bool KC>
__device__ __forceinline__ void tile_store_bf16(const float4 (&r)[4], unsigned short* __restrict__ S, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + 256 * it;
    if (KC) {
      const int row = c >> 3, k = (c & 7) << 2;
      uint2 v = make_uint2(pack_bf16(r[it].x, r[it].y), pack_bf16(r[it].z, r[it].w));
      *reinterpret_cast<uint2*>(&S[row * HROW + k]) = v;
    } else {
      const int k = c >> 5, row = (c & 31) << 2;
      S[(row + 0) * HROW + k] = __builtin_bit_cast(unsigned short, (__bf16)r[it].x);
      S[(row + 1) * HROW + k] = __builtin_bit_cast(unsigned short, (__bf16)r[it].y);
      S[(row + 2) * HROW + k] = __builtin_bit_cast(unsigned short, (__bf16)r[it].z);
      S[(row + 3) * HROW + k] = __builtin_bit_cast(unsigned short, (__bf16)r[it].w);
    }
  }
}

template <bool AKC, bool BKC, bool FAST>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * HROW];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * HROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int bx = (int)(lid % g.gx);
  const int by = (int)((lid / g.gx) % g.gy);
  const int bz = (int)(lid / ((int64_t)g.gx * g.gy));
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
  const int64_t kbeg = (int64_t)bz * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  const int64_t nk = (kend - kbeg + BK - 1) / BK;
  const int lr = lane & 31, lk = lane >> 5;
  if (FAST) {  // all tiles interior: unconditional loads issued above the MFMA block (see gemm_f32_kernel)
    tile_load_fast<AKC>(ra, g.A, g.lda, m0, kbeg, tid);
    tile_load_fast<BKC>(rb, g.B, g.ldb, n0, kbeg, tid);
  } else if (nk > 0) {
    tile_load<AKC>(ra, g.A, g.lda, g.M, kend, m0, kbeg, g.vecA, tid);
    tile_load<BKC>(rb, g.B, g.ldb, g.N, kend, n0, kbeg, g.vecB, tid);
  }
  if (nk > 0) {
    tile_store_bf16<AKC>(ra, As[0], tid);
    tile_store_bf16<BKC>(rb, Bs[0], tid);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (FAST) {
      const int64_t kn = kbeg + (kt + 1 < nk ? kt + 1 : kt) * BK;
      tile_load_fast<AKC>(ra, g.A, g.lda, m0, kn, tid);
      tile_load_fast<BKC>(rb, g.B, g.ldb, n0, kn, tid);
      __builtin_amdgcn_sched_barrier(0);
    } else if (kt + 1 < nk) {
      tile_load<AKC>(ra, g.A, g.lda, g.M, kend, m0, kbeg + (kt + 1) * BK, g.vecA, tid);
      tile_load<BKC>(rb, g.B, g.ldb, g.N, kend, n0, kbeg + (kt + 1) * BK, g.vecB, tid);
    }
    const unsigned short* as = As[cur] + (wm * 64 + lr) * HROW + lk * 8;
    const unsigned short* bs = Bs[cur] + (wn * 64 + lr) * HROW + lk * 8;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(as + ks * 16);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(as + 32 * HROW + ks * 16);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bs + ks * 16);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bs + 32 * HROW + ks * 16);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (FAST || kt + 1 < nk) {
      tile_store_bf16<AKC>(ra, As[cur ^ 1], tid);
      tile_store_bf16<BKC>(rb, Bs[cur ^ 1], tid);
    }
    __syncthreads();
  }
  gemm_epilogue(g, acc, m0, n0, wm, wn, lr, lk, bz);
  if (g.bn_part) gemm_tile_bn_stats(g, acc, reinterpret_cast<float*>(&As[0][0]), m0, n0, wm, wn, lr, lk, by);
}

// ---- bf16-RESIDENT variant (the MLP's use_amp path): operands already live in HBM as bf16, so a 128 x 64 operand
// tile is 16 KB instead of the 32 KB (read as fp32) of a 32-deep tile above and needs no conversion while staging.
//   TN = false: A stored (M,K), B stored (N,K), k contiguous: LDS images [m|n][k] (rows padded to 72 bf16 = 144 B, so
//               the 16 lanes of a ds_read_b128 group hit 16 distinct 16-B slots); a lane's fragment = one ds_read_b128.
//   TN = true : A stored (K,M), B stored (K,N) (wgrad: dW = dy^T x, both operands "row = sample"): the tile is copied
//               into LDS as it lies ([k][m|n], 16-B chunks along m) and the MFMA fragments — 8 consecutive k of one
//               m — come out of ds_read_b64_tr_b16, the gfx950 transposing LDS read: per 16-lane group it reads a block
//               of 4 rows (k) x 16 columns (m) and hands lane i column i.  Row stride 160 bf16 = 80 dwords = 16 (mod 64
//               banks): the four rows of a block cover the 64 banks exactly once.
// Interior tiles only (M, N % 128 == 0, K % 64 == 0 per split, 16-B aligned rows): checked on the host.
constexpr int BK2 = 64;
constexpr int NT_ROW = 72;    // bf16 per LDS row of an [m][k] image (64 + 8 pad); [k][m] images: rows of (tile width + 32)
using i16x4 = __attribute__((ext_vector_type(4))) short;
using i32x4 = __attribute__((ext_vector_type(4))) int;

struct Gemm16Args {
  const unsigned short* A;
  const unsigned short* B;
  float* C;
  const float* bias;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  float alpha, beta;
  int64_t k_per_split;
  float* slabs;
  int gx, gy, splits;
  float* bn_part;
  unsigned short* C16;
};

// one ROWS x 64 (NT) or 64 x ROWS (TN) bf16 tile = ROWS * 8 16-byte chunks, NCH = ROWS * 8 / THREADS per thread;
// TROW = LDS row stride of the TN image (ROWS + 32: 16 dwords mod 64 banks for ROWS = 128 and 256)
template <bool TN, int ROWS, int THREADS>
__device__ __forceinline__ void tile16_load(i32x4 (&r)[ROWS * 8 / THREADS], const unsigned short* __restrict__ P,
                                            int64_t ld, int64_t row0, int64_t k0, int tid) {
  constexpr int CPR = ROWS / 8;  // chunks per k-row of the TN form
#pragma unroll
  for (int it = 0; it < ROWS * 8 / THREADS; ++it) {
    const int c = tid + THREADS * it;
    const unsigned short* p = TN ? P + (k0 + c / CPR) * ld + row0 + ((c % CPR) << 3)
                                 : P + (row0 + (c >> 3)) * ld + k0 + ((c & 7) << 3);   // 8 chunks per row of 64 k
    r[it] = *reinterpret_cast<const i32x4*>(p);
  }
}
template <bool TN, int ROWS, int THREADS>
__device__ __forceinline__ void tile16_store(const i32x4 (&r)[ROWS * 8 / THREADS], unsigned short* __restrict__ S,
                                             int tid) {
  constexpr int CPR = ROWS / 8, TROW = ROWS + 32;
#pragma unroll
  for (int it = 0; it < ROWS * 8 / THREADS; ++it) {
    const int c = tid + THREADS * it;
    unsigned short* d = TN ? S + (c / CPR) * TROW + ((c % CPR) << 3) : S + (c >> 3) * NT_ROW + ((c & 7) << 3);
    *reinterpret_cast<i32x4*>(d) = r[it];
  }
}

// BMT = 128: 4 waves (2 x 2), two workgroups per CU.  BMT = 256: 8 waves (4 x 2), one workgroup per CU — the A tile is
// shared by twice as many MFMAs, which matters because with 128 x 128 tiles the operand stream from L2 (~30 B/clk/CU),
// not the matrix cores, bounds the kernel.
template <bool TN, int BMT, int BNT = BN>
__global__ __launch_bounds__(BMT * BNT / 64) void gemm_bf16in_kernel(const Gemm16Args g) {
  constexpr int THREADS = BMT * BNT / 64;  // one wave per 64 x 64 block of the tile
  constexpr int WN = BNT / 64;
  constexpr int TROW_A = BMT + 32, TROW_B = BNT + 32;  // TN image row strides (bf16): 16 dwords mod 64 banks
  constexpr int IMG_A = TN ? BK2 * TROW_A : BMT * NT_ROW;
  constexpr int IMG_B = TN ? BK2 * TROW_B : BNT * NT_ROW;
  __shared__ __attribute__((aligned(16))) unsigned short As[2][IMG_A];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][IMG_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);  // XCD-aware tile order (see gemm_f32_kernel)
  const int bx = (int)(lid % g.gx);
  const int by = (int)((lid / g.gx) % g.gy);
  const int bz = (int)(lid / ((int64_t)g.gx * g.gy));
  const int64_t m0 = (int64_t)by * BMT, n0 = (int64_t)bx * BNT;
  const int64_t kbeg = (int64_t)bz * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int64_t nk = (kend - kbeg) / BK2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  // TN fragment addressing: 16-lane group gq = lane / 16 reads m-columns (gq & 1) * 16 .. +15 and k-rows
  // (gq >> 1) * 8 .. +7 in two blocks of 4 rows; lane 4q + p of the group supplies row q, columns 4p .. 4p+3
  const int gq = lane >> 4, li = lane & 15;
  const int tr_a = (((gq >> 1) * 8 + (li >> 2)) * TROW_A + (gq & 1) * 16 + ((li & 3) << 2));
  const int tr_b = (((gq >> 1) * 8 + (li >> 2)) * TROW_B + (gq & 1) * 16 + ((li & 3) << 2));

  i32x4 ra[BMT * 8 / THREADS], rb[BNT * 8 / THREADS];
  tile16_load<TN, BMT, THREADS>(ra, g.A, g.lda, m0, kbeg, tid);
  tile16_load<TN, BNT, THREADS>(rb, g.B, g.ldb, n0, kbeg, tid);
  tile16_store<TN, BMT, THREADS>(ra, As[0], tid);
  tile16_store<TN, BNT, THREADS>(rb, Bs[0], tid);
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    const int64_t kn = kbeg + (kt + 1 < nk ? kt + 1 : kt) * BK2;
    tile16_load<TN, BMT, THREADS>(ra, g.A, g.lda, m0, kn, tid);
    tile16_load<TN, BNT, THREADS>(rb, g.B, g.ldb, n0, kn, tid);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      bf16x8 a0, a1, b0, b1;
      if (TN) {
        using lds_v4 = __attribute__((address_space(3))) i16x4;
        const unsigned short* ab = As[cur] + ks * 16 * TROW_A + wm * 64 + tr_a;
        const unsigned short* bb = Bs[cur] + ks * 16 * TROW_B + wn * 64 + tr_b;
        i16x4 t[8];
        t[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab));
        t[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab + 4 * TROW_A));
        t[2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab + 32));
        t[3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab + 32 + 4 * TROW_A));
        t[4] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(bb));
        t[5] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(bb + 4 * TROW_B));
        t[6] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(bb + 32));
        t[7] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(bb + 32 + 4 * TROW_B));
        a0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t[0], t[1], 0, 1, 2, 3, 4, 5, 6, 7));
        a1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t[2], t[3], 0, 1, 2, 3, 4, 5, 6, 7));
        b0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t[4], t[5], 0, 1, 2, 3, 4, 5, 6, 7));
        b1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t[6], t[7], 0, 1, 2, 3, 4, 5, 6, 7));
      } else {
        const unsigned short* as = As[cur] + (wm * 64 + lr) * NT_ROW + ks * 16 + lk * 8;
        const unsigned short* bs = Bs[cur] + (wn * 64 + lr) * NT_ROW + ks * 16 + lk * 8;
        a0 = *reinterpret_cast<const bf16x8*>(as);
        a1 = *reinterpret_cast<const bf16x8*>(as + 32 * NT_ROW);
        b0 = *reinterpret_cast<const bf16x8*>(bs);
        b1 = *reinterpret_cast<const bf16x8*>(bs + 32 * NT_ROW);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    tile16_store<TN, BMT, THREADS>(ra, As[cur ^ 1], tid);
    tile16_store<TN, BNT, THREADS>(rb, Bs[cur ^ 1], tid);
    __syncthreads();
  }
  GemmArgs e = {};
  e.C = g.C; e.bias = g.bias; e.M = g.M; e.N = g.N; e.K = g.K; e.ldc = g.ldc; e.alpha = g.alpha; e.beta = g.beta;
  e.slabs = g.slabs; e.splits = g.splits; e.bn_part = g.bn_part; e.C16 = g.C16;
  if (g.C16) gemm_epilogue<true>(e, acc, m0, n0, wm, wn, lr, lk, bz);
  else gemm_epilogue<false>(e, acc, m0, n0, wm, wn, lr, lk, bz);
  if (g.bn_part)
    gemm_tile_bn_stats(e, acc, reinterpret_cast<float*>(&As[0][0]), m0, n0, wm, wn, lr, lk, by, BMT / BM, WN);
}

// ---- NT form at 256 x 256 x 64 tiles filled by the LDS-DMA (global_load_lds_dwordx4): forward and input-gradient GEMMs
// of the bf16-resident MLP step.  8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 accumulators: a
// k-step of 16 reads 6 fragments for 8 MFMAs (the 16-wave kernel above: 4 for 4 — at 1 KB of LDS reads per MFMA the LDS
// array, not the matrix cores, set its pace).  No staging registers, no ds_write pass: each wave-instruction moves
// 8 rows x 128 B (64 k of bf16) straight into LDS, lane-linear; the bank-conflict swizzle (16-B slot ^= (row >> 1) & 7,
// which spreads every 16-lane group of a ds_read_b128 over the 16 slots of the 256-B bank row) is applied to the per-lane
// SOURCE address and again on the fragment reads.  One barrier per k-tile: the loads of tile t+1 are issued during tile t's
// k-steps (two DMA pieces per step) and awaited (vmcnt(0)) at the next barrier.  Measured against the 16-wave kernel at
// the c5 shapes (65 536 rows): 1280 -> 1024: 182 us (944 TFLOP/s) against 221; 1024 -> 512: 73 against 96;
// 512 -> 256: 24 against 37 (tools/micro/gemm16_bench.hip).
// Epilogue: alpha * acc + bias -> fp32 or bf16 C; BatchNorm partial statistics per 128-row chunk = per wave row (every
// column's 128 rows live in ONE wave: 4 accumulators x 16 registers x 2 lane halves — no LDS, no barrier).
constexpr int G3_TILE = 256 * BK2 * 2;  // bytes of one operand tile
typedef const __attribute__((address_space(1))) void* g3_gptr;
typedef __attribute__((address_space(3))) void* g3_lptr;

__device__ __forceinline__ void g3_stage(const unsigned short* __restrict__ P, int64_t ld, int64_t row0, int64_t k0,
                                         char* lds_tile, int wave, int lane) {
  const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // 32 pieces of 8 rows per tile, 4 per wave
    const int piece = wave * 4 + q;
    const int row = piece * 8 + r8;
    const unsigned short* src = P + (row0 + row) * ld + k0 + ((slot ^ ((row >> 1) & 7)) << 3);
    __builtin_amdgcn_global_load_lds((g3_gptr)src, (g3_lptr)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

// piece p (0-3: A, 4-7: B) of a wave's 8 DMA instructions for one k-tile
__device__ __forceinline__ void g3_piece(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                         int64_t lda, int64_t ldb, int64_t m0, int64_t n0, int64_t k0, char* lds_buf,
                                         int wave, int lane, int p) {
  const bool isb = p >= 4;
  const int piece = wave * 4 + (p & 3);
  const int row = piece * 8 + (lane >> 3);
  const unsigned short* src = (isb ? B + (n0 + row) * ldb : A + (m0 + row) * lda) + k0 +
                              (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  __builtin_amdgcn_global_load_lds((g3_gptr)src, (g3_lptr)(lds_buf + (isb ? G3_TILE : 0) + piece * 1024), 16, 0, 0);
}

// Epilogue of the 256 x 256 bf16 NT kernels (gemm16_nt_glds_kernel, gemm16_gather_nt_kernel): alpha * acc + bias, fp32 or
// bf16 rows through the wave's own 16 KB of the dead operand buffers, BatchNorm partials per 128-row chunk.
template <bool OUT16>
__device__ __forceinline__ void g3_epilogue(const Gemm16Args& g, f32x16 (&acc)[4][2], char* lds, int64_t m0, int64_t n0,
                                            int by, int wave, int lane) {
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lk = lane >> 5;
  // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) — a lane holds
  // ONE column of 16 rows, so storing from the accumulators would issue 128 stores of 64-byte row pieces per wave.  The
  // operand buffers are dead: each wave transposes its tile through its own 16 KB of LDS, 64 rows at a time, and leaves
  // through 16-byte stores of whole 128-byte (bf16) / 256-byte (fp32) row segments.
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bv[j] = g.bias ? g.bias[n0 + wn * 64 + j * 32 + lr] : 0.f;
  __syncthreads();  // every wave has read the last k-tile
  float* stage = reinterpret_cast<float*>(lds + wave * 16384);  // [64 rows][64 columns] fp32, private to the wave
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = g.alpha * acc[2 * h + ii][j][r] + bv[j];
          acc[2 * h + ii][j][r] = v;
          stage[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = v;
        }
    const int64_t row_h = m0 + wm * 128 + h * 64, col_w = n0 + wn * 64;
    if (OUT16) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // 8 rows per wave-instruction: 8 lanes x 8 columns each
        const int row = q * 8 + (lane >> 3), cg = lane & 7;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 64 + cg * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * 64 + cg * 8 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (__bf16)lo[e];
          o[4 + e] = (__bf16)hi[e];
        }
        *reinterpret_cast<bf16x8*>(g.C16 + (row_h + row) * g.ldc + col_w + cg * 8) = o;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // 4 rows per wave-instruction: 16 lanes x 4 columns each
        const int row = q * 4 + (lane >> 4), cg = lane & 15;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cg * 4);
        *reinterpret_cast<f32x4*>(g.C + (row_h + row) * g.ldc + col_w + cg * 4) = v4;
      }
    }
  }
  if (g.bn_part) {  // mean and sum of squared deviations of the stored values over this wave's 128 rows, per column
    float* o = g.bn_part + (int64_t)(by * 2 + wm) * 2 * g.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.0f / 128.0f);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[i][j][r] - mean;
          q += d * d;
        }
      q += __shfl_xor(q, 32, 64);
      if (lk == 0) {
        const int64_t col = n0 + wn * 64 + j * 32 + lr;
        o[col] = mean;
        o[g.N + col] = q;
      }
    }
  }
}

template <bool OUT16>
__global__ __launch_bounds__(512) void gemm16_nt_glds_kernel(const Gemm16Args g) {
  __shared__ __attribute__((aligned(1024))) char lds[4 * G3_TILE];  // [buffer][A | B], the only LDS object
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);  // XCD-aware tile order (see gemm_f32_kernel)
  const int bx = (int)(lid % g.gx), by = (int)(lid / g.gx);
  const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;
  const int nk = (int)(g.K / BK2);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lr = lane & 31, lk = lane >> 5;
  g3_stage(g.A, g.lda, m0, 0, lds, wave, lane);
  g3_stage(g.B, g.ldb, n0, 0, lds + G3_TILE, wave, lane);
  const int f = (lr >> 1) & 7;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // (drains this wave's LDS-DMA: vmcnt(0)) tile kt is in LDS, tile kt - 1 has been read by everyone
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    char* nxt = lds + (cur ^ 1) * 2 * G3_TILE;
    const int64_t kn = (int64_t)(kt + 1) * BK2;
    const char* ta = lds + cur * 2 * G3_TILE + (wm * 128 + lr) * 128;
    const char* tb = lds + cur * 2 * G3_TILE + G3_TILE + (wn * 64 + lr) * 128;
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      const int sw = ((ks * 2 + lk) ^ f) << 4;
      bf16x8 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ta + i * 32 * 128 + sw);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + sw);
      // the next tile's 8 LDS-DMA pieces of this wave go out two per k-step, behind the step's fragment reads: all 8 right
      // after the barrier keep every wave's first reads and MFMAs waiting behind ~1000 cycles of DMA issue (fwd 1280 ->
      // 1024: 181 -> 163 us; tools/micro/gemm16_bench.hip)
      if (more) {
        g3_piece(g.A, g.B, g.lda, g.ldb, m0, n0, kn, nxt, wave, lane, 2 * ks);
        g3_piece(g.A, g.B, g.lda, g.ldb, m0, n0, kn, nxt, wave, lane, 2 * ks + 1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  g3_epilogue<OUT16>(g, acc, lds, m0, n0, by, wave, lane);
}

// ---- MLP layer 0 with the embedding gather INSIDE the GEMM's A-operand load (the "concat-GEMM" of BASELINE.json's
// north_star; reference collaborative/mlp.py:93-105: three / five nn.Embedding gathers, two torch.cat, then fcs[0]).
// Row r of the virtual A = x0 is [user[u] | item[i] | meta_1[..] ...]: a k-tile lies inside ONE field (D a multiple of the
// tile depth), so the tile's rows come straight from that field's table, addressed by the ids of the workgroup's 256
// rows (validated once, kept in LDS).  No gather kernel, no read of a materialised x0 by this GEMM; the x0 image the
// weight-gradient GEMM reads later is written by the column-block-0 workgroups from what they staged anyway.
constexpr int GATHER_FIELDS = 2 + TRS_MAX_META;
struct GatherSrc {
  const float* tab[GATHER_FIELDS];
  const int32_t* ids[2][GATHER_FIELDS];  // [pass][field]: int32 ids of the pass's B rows
  int32_t stride[GATHER_FIELDS];         // id stride (metadata ids are (B, M) row-major)
  int64_t n_rows[GATHER_FIELDS];
  int64_t B;
  int32_t D, F;
  int32_t* err;
};

// ids of the workgroup's 256 rows (row r of the stacked passes: pass = r >= B) -> LDS, out-of-range ids clamped to row 0
// and reported; table base pointers -> LDS (a field index that changes with the k-tile would otherwise index the kernel
// argument struct dynamically)
typedef const __attribute__((address_space(1))) float* gfloat_ptr;  // global loads, not flat ones (a flat load also
// counts on lgkmcnt and is drained by every barrier's LDS wait)
__device__ __forceinline__ void gather_ids_to_lds(const GatherSrc& S, int64_t m0, int32_t* s_ids, gfloat_ptr* s_tab,
                                                  int tid) {
  for (int i = tid; i < S.F * 256; i += 512) {
    const int f = i >> 8, row = i & 255;
    const int64_t r = m0 + row;
    const int pass = r >= S.B;
    const int64_t t = pass ? r - S.B : r;
    int32_t id = S.ids[pass][f][t * S.stride[f]];
    if ((uint64_t)(int64_t)id >= (uint64_t)S.n_rows[f]) {
      id = 0;
      if (S.err) atomicOr(S.err, 1);
    }
    s_ids[i] = id;
  }
  if (tid < S.F) s_tab[tid] = (gfloat_ptr)S.tab[tid];
}

// bf16-resident form: B (the bf16 weight image) by LDS-DMA as in gemm16_nt_glds_kernel; A register-staged — a thread
// loads the 32 bytes of fp32 behind each of its four 16-byte bf16 chunks of the next k-tile at the top of the current
// tile (in flight under the MFMAs), rounds them (RNE: v_cvt_pk_bf16_f32, the rounding of the gather kernel it replaces)
// and writes them into the swizzled LDS image the fragment reads expect — and, in column block 0, into the x0 image.
template <bool OUT16>
__global__ __launch_bounds__(512) void gemm16_gather_nt_kernel(const Gemm16Args g, const GatherSrc S,
                                                              unsigned short* __restrict__ x16, int64_t ldx) {
  // ONE LDS object (tiles | ids | table pointers): with several, the compiler can no longer tell the LDS-DMA's target
  // from the other LDS accesses and puts s_waitcnt vmcnt(0) behind every DMA piece (each piece then waits for its own
  // landing, and for the A loads in flight)
  __shared__ __attribute__((aligned(1024))) char lds[4 * G3_TILE + GATHER_FIELDS * 256 * 4 + GATHER_FIELDS * 8];
  int32_t* s_ids = reinterpret_cast<int32_t*>(lds + 4 * G3_TILE);
  gfloat_ptr* s_tab = reinterpret_cast<gfloat_ptr*>(lds + 4 * G3_TILE + GATHER_FIELDS * 256 * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);  // XCD-aware tile order (see gemm_f32_kernel)
  const int bx = (int)(lid % g.gx), by = (int)(lid / g.gx);
  const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;
  const int nk = (int)(g.K / BK2);
  gather_ids_to_lds(S, m0, s_ids, s_tab, tid);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lr = lane & 31, lk = lane >> 5;
  const int D = S.D;
  const bool image = x16 != nullptr && bx == 0;
  // chunk c = tid + 512 * it: row = c >> 3 = (tid >> 3) + 64 * it, 16-byte bf16 chunk kc = c & 7 = tid & 7 of the row's 64 k
  const int kc = tid & 7, r0 = tid >> 3;
  f32x4 ra[4][2];
  auto a_issue = [&](int kt) {
    const int k0 = kt * BK2, f = k0 / D, off = k0 - f * D + kc * 8;
    const gfloat_ptr tab = s_tab[f];
    typedef const __attribute__((address_space(1))) f32x4* gvec_ptr;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const gfloat_ptr p = tab + (int64_t)s_ids[f * 256 + r0 + 64 * it] * D + off;
      ra[it][0] = *(gvec_ptr)p;
      ra[it][1] = *(gvec_ptr)(p + 4);
    }
  };
  auto a_commit = [&](char* tile, int kt, bool img) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = r0 + 64 * it;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (__bf16)ra[it][0][e];
        v[4 + e] = (__bf16)ra[it][1][e];
      }
      // (inline asm: a ds_write the compiler can see is ordered behind the LDS-DMA pieces just issued for the same buffer —
      // it cannot tell the A half from the B half — i.e. s_waitcnt vmcnt(0) in front of the commit, which exposes the
      // DMA's whole latency every k-tile; the barrier's lgkmcnt(0) covers this write)
      const uint32_t la = (uint32_t)(uintptr_t)(g3_lptr)(tile + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(__builtin_bit_cast(i32x4, v)) : "memory");
      if (img) *reinterpret_cast<bf16x8*>(x16 + (m0 + row) * ldx + (int64_t)kt * BK2 + kc * 8) = v;
    }
  };
  __syncthreads();  // ids and table pointers are in LDS
  a_issue(0);
  g3_stage(g.B, g.ldb, n0, 0, lds + G3_TILE, wave, lane);
  a_commit(lds, 0, image);
  // The loads of tile t+1 are issued right after tile t was committed — a whole iteration (barrier, MFMA block) before
  // their own commit.  Issued at the top of tile t instead (one MFMA block, ~0.9 us, ahead) they stalled every k-tile
  // by ~1.2 us: a random table row from HBM takes longer than that under load (c5's first layer: 290 against 195 us for
  // the LDS-DMA kernel on a materialised x0).  Pulling tile t+2 into L2 by 4-byte LDS-DMAs into a sink was slower still
  // (+60 us per step: the DMA issue slots).
  a_issue(nk > 1 ? 1 : 0);
  const int fsw = (lr >> 1) & 7;
  for (int kt = 0; kt < nk; ++kt) {
    // Everything older than the 8 A loads issued last (this wave's LDS-DMA pieces and image stores of the previous
    // iteration) has landed; the A loads stay in flight across the barrier.  Explicit: the compiler orders an LDS-DMA
    // only against THIS wave's later LDS accesses, the barrier needs it for the other waves' reads.
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();  // tile kt is in LDS (A: ds_write, B: LDS-DMA), tile kt - 1 has been read by everyone
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    char* nxt = lds + (cur ^ 1) * 2 * G3_TILE;
    const int64_t kn = (int64_t)(more ? kt + 1 : kt) * BK2;
    const char* ta = lds + cur * 2 * G3_TILE + (wm * 128 + lr) * 128;
    const char* tb = lds + cur * 2 * G3_TILE + G3_TILE + (wn * 64 + lr) * 128;
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      const int sw = ((ks * 2 + lk) ^ fsw) << 4;
      bf16x8 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ta + i * 32 * 128 + sw);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tb + j * 32 * 128 + sw);
      // B: its four DMA pieces in the first two k-steps — the commit below waits for EVERYTHING in flight (the compiler
      // cannot count the A loads across the loop's back edge and emits vmcnt(0)), so the pieces should have landed by
      // then (the weights come from L2).  Nothing in this loop sits under a branch: at the join of a conditional block
      // the compiler waits for every load in flight too, which would drain the A loads at the first k-step; the last
      // iteration re-stages its own tile into the dead buffer instead
      if (ks < 2) {
        g3_piece(g.A, g.B, g.lda, g.ldb, m0, n0, kn, nxt, wave, lane, 4 + 2 * ks);
        g3_piece(g.A, g.B, g.lda, g.ldb, m0, n0, kn, nxt, wave, lane, 5 + 2 * ks);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the conversion of the staged A rows stays behind the MFMA block ...
    a_commit(nxt, more ? kt + 1 : kt, image && more);  // (last iteration: its own tile again, into the dead buffer)
    __builtin_amdgcn_sched_barrier(0);  // ... and the next loads behind the commit that frees their registers)
    a_issue(kt + 2 < nk ? kt + 2 : nk - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  g3_epilogue<OUT16>(g, acc, lds, m0, n0, by, wave, lane);
}

// ---- fp32 NT form on the LDS-DMA structure: C(M,N) = alpha * A(M,K) B(N,K)^T (+ bias), both operands k-contiguous
// (forward y = x W^T; input gradient dx = dy W through a transposed copy of W).  A tile row = 32 floats = 128 bytes — the
// byte geometry of the bf16 kernel above, so the DMA pieces, the swizzle and the epilogue are the same; a fragment read
// is one ds_read_b128 = four consecutive k of a row (both lane halves read the same 16 bytes), which feeds two
// v_mfma_f32_32x32x2_f32 steps (lane half lk takes element lk, then 2 + lk).  256 x BNT x 32 tiles, 8 waves: BNT = 256
// as 2 x 4 waves of 128 x 64 (BatchNorm partials wave-local), BNT = 128 as 4 x 2 waves of 64 x 64 (no statistics).
// Per k-tile a wave issues 128 (64) MFMAs of 64 cycles against one 64 (48) KB tile of DMA: the matrix pipe, not the
// load path, sets the pace (the register-staged 128 x 128 kernel: 102 TFLOP/s at c3's shapes).
__device__ __forceinline__ void g32_piece(const float* __restrict__ P, int64_t ld, int64_t row0, int64_t k0,
                                          char* lds_tile, int piece, int lane) {
  const int row = piece * 8 + (lane >> 3);
  const float* src = P + (row0 + row) * ld + k0 + (((lane & 7) ^ ((row >> 1) & 7)) << 2);
  __builtin_amdgcn_global_load_lds((g3_gptr)src, (g3_lptr)(lds_tile + piece * 1024), 16, 0, 0);
}

// GATHER: A = the concatenated embedding rows of MLP layer 0, fetched by id from the tables (GatherSrc above) — the DMA
// piece of a tile row takes its source address from the row's id instead of a row stride; the column-block-0 workgroups
// copy every landed A tile to the fp32 x0 image the weight-gradient GEMM reads (four ds_read_b128 + 16-byte stores per
// wave and k-tile).
__device__ __forceinline__ void g32_piece_gather(const gfloat_ptr* s_tab, const int32_t* s_ids, int D, int k0,
                                                 char* lds_tile, int piece, int lane) {
  const int row = piece * 8 + (lane >> 3);
  const int f = k0 / D, off = k0 - f * D;
  const gfloat_ptr src = s_tab[f] + (int64_t)s_ids[f * 256 + row] * D + off + (((lane & 7) ^ ((row >> 1) & 7)) << 2);
  __builtin_amdgcn_global_load_lds((g3_gptr)src, (g3_lptr)(lds_tile + piece * 1024), 16, 0, 0);
}

template <int BNT, bool GATHER = false>
__global__ __launch_bounds__(512) void gemm32_nt_glds_kernel(const GemmArgs g, const GatherSrc S, float* __restrict__ ximg,
                                                            int64_t ldx) {
  constexpr int WN = BNT / 64, WM = 8 / WN, TM = 256 / WM, MI = TM / 32;  // 2 x 4 of 128 x 64 | 4 x 2 of 64 x 64
  constexpr int TILE_A = 256 * 128, TILE_B = BNT * 128, STAGE = TILE_A + TILE_B;
  constexpr int PA = 32 / 8, PB = (BNT / 8) / 8;  // DMA pieces per wave and tile: A 4, B 4 | 2
  constexpr int LDS_TILES = 2 * STAGE > 8 * 16384 ? 2 * STAGE : 8 * 16384;
  // (ONE LDS object: see gemm16_gather_nt_kernel)
  __shared__ __attribute__((aligned(1024))) char lds[LDS_TILES + (GATHER ? GATHER_FIELDS * 256 * 4 + GATHER_FIELDS * 8 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int64_t lid = blockIdx.x;
  const int64_t nwg = gridDim.x;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int bx = (int)(lid % g.gx), by = (int)(lid / g.gx);
  const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * BNT;
  const int nk = (int)(g.K / 32);
  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lr = lane & 31, lk = lane >> 5;
  const int f = (lr >> 1) & 7;
  int32_t* s_ids = reinterpret_cast<int32_t*>(lds + LDS_TILES);
  gfloat_ptr* s_tab = reinterpret_cast<gfloat_ptr*>(lds + LDS_TILES + GATHER_FIELDS * 256 * 4);
  if (GATHER) {
    gather_ids_to_lds(S, m0, s_ids, s_tab, tid);
    __syncthreads();
  }
  const bool image = GATHER && ximg != nullptr && bx == 0;
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    if (GATHER) g32_piece_gather(s_tab, s_ids, S.D, 0, lds, wave * PA + q, lane);
    else g32_piece(g.A, g.lda, m0, 0, lds, wave * PA + q, lane);
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) g32_piece(g.B, g.ldb, n0, 0, lds + TILE_A, wave * PB + q, lane);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    char* nxt = lds + (cur ^ 1) * STAGE;
    const int64_t kn = (int64_t)(kt + 1) * 32;
    const char* ta = lds + cur * STAGE + (wm * TM + lr) * 128;
    const char* tb = lds + cur * STAGE + TILE_A + (wn * 64 + lr) * 128;
    if (image) {  // the landed A tile -> x0 image: LDS slot (lane & 7) of a row holds source chunk slot ^ ((row >> 1) & 7)
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int piece = wave * PA + q, row = piece * 8 + (lane >> 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + cur * STAGE + piece * 1024 + lane * 16);
        *reinterpret_cast<f32x4*>(ximg + (m0 + row) * ldx + (int64_t)kt * 32 + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) = v;
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // 16-byte chunk = 4 k
      const int sw = (c ^ f) << 4;
      f32x4 a[MI], b[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(ta + i * 32 * 128 + sw);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(tb + j * 32 * 128 + sw);
      if (more) {  // the next tile's DMA pieces, one or two per chunk step, behind the fragment reads
        if (c < PA) {
          if (GATHER) g32_piece_gather(s_tab, s_ids, S.D, (int)kn, nxt, wave * PA + c, lane);
          else g32_piece(g.A, g.lda, m0, kn, nxt, wave * PA + c, lane);
        } else if (c - PA < PB) g32_piece(g.B, g.ldb, n0, kn, nxt + TILE_A, wave * PB + (c - PA), lane);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float av = lk ? a[i][2 * h + 1] : a[i][2 * h];
            const float bv = lk ? b[j][2 * h + 1] : b[j][2 * h];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
      }
    }
  }
  // epilogue (as gemm16_nt_glds_kernel): alpha * acc + bias through the wave's 16 KB of LDS, 16-byte stores
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bv[j] = g.bias ? g.bias[n0 + wn * 64 + j * 32 + lr] : 0.f;
  __syncthreads();
  float* stage = reinterpret_cast<float*>(lds + wave * 16384);
#pragma unroll
  for (int h = 0; h < MI / 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = g.alpha * acc[2 * h + ii][j][r] + bv[j];
          acc[2 * h + ii][j][r] = v;
          stage[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = v;
        }
    const int64_t row_h = m0 + wm * TM + h * 64, col_w = n0 + wn * 64;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = q * 4 + (lane >> 4), cg = lane & 15;
      *reinterpret_cast<f32x4*>(g.C + (row_h + row) * g.ldc + col_w + cg * 4) =
          *reinterpret_cast<const f32x4*>(stage + row * 64 + cg * 4);
    }
  }
  if (MI == 4 && g.bn_part) {  // (BNT = 256) statistics of the stored values over this wave's 128 rows, per column
    float* o = g.bn_part + (int64_t)(by * 2 + wm) * 2 * g.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s_ = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_ += acc[i][j][r];
      s_ += __shfl_xor(s_, 32, 64);
      const float mean = s_ * (1.0f / 128.0f);
      float q_ = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[i][j][r] - mean;
          q_ += d * d;
        }
      q_ += __shfl_xor(q_, 32, 64);
      if (lk == 0) {
        const int64_t col = n0 + wn * 64 + j * 32 + lr;
        o[col] = mean;
        o[g.N + col] = q_;
      }
    }
  }
}

// fp32 (rows, cols) -> bf16 copy (same layout) and, optionally, the transposed bf16 copy (cols, rows): the per-step
// refresh of the MLP's weight images (tiny: the weights, not the activations).
__device__ __forceinline__ void f32_to_bf16_tile(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld,
                                                 unsigned short* __restrict__ dst, unsigned short* __restrict__ dst_t,
                                                 int64_t tile_x, int64_t tile_y) {
  __shared__ float tile[32][33];
  const int64_t r0 = tile_y * 32, c0 = tile_x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = r0 + j, c = c0 + tx;
    const float v = (r < rows && c < cols) ? src[r * ld + c] : 0.f;
    tile[j][tx] = v;
    if (dst && r < rows && c < cols) dst[r * cols + c] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
  __syncthreads();
  if (dst_t)
    for (int j = ty; j < 32; j += 8) {
      const int64_t c = c0 + j, r = r0 + tx;
      if (r < rows && c < cols) dst_t[c * rows + r] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][j]);
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, int64_t rows, int64_t cols,
                                                         int64_t ld, unsigned short* __restrict__ dst,
                                                         unsigned short* __restrict__ dst_t) {
  f32_to_bf16_tile(src, rows, cols, ld, dst, dst_t, blockIdx.x, blockIdx.y);
}

// the same for up to TRS_IMG_MAX matrices in one launch (the weight images of every layer of the MLP): workgroup b serves
// tile b - first[k] of matrix k
constexpr int TRS_IMG_MAX = 8;
struct ImgArgs {
  const float* src[TRS_IMG_MAX];
  unsigned short* dst[TRS_IMG_MAX];
  unsigned short* dst_t[TRS_IMG_MAX];
  int64_t rows[TRS_IMG_MAX], cols[TRS_IMG_MAX], ld[TRS_IMG_MAX];
  int64_t first[TRS_IMG_MAX + 1];  // first tile of matrix k; first[n] = total
  int n;
};
__global__ __launch_bounds__(256) void f32_to_bf16_multi_kernel(const ImgArgs a) {
  int k = 0;
  while (k + 1 < a.n && (int64_t)blockIdx.x >= a.first[k + 1]) ++k;
  const int64_t t = (int64_t)blockIdx.x - a.first[k];
  const int64_t tiles_x = (a.cols[k] + 31) / 32;
  f32_to_bf16_tile(a.src[k], a.rows[k], a.cols[k], a.ld[k], a.dst[k], a.dst_t[k], t % tiles_x, t / tiles_x);
}

// C = alpha * sum_z slabs[z] + beta * C (+ bias): the launch-boundary reduce of the split-K partial slabs, fixed
// summation order (bitwise reproducible).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int64_t M,
                                                           int64_t N, float* __restrict__ C, int64_t ldc,
                                                           float alpha, float beta, const float* __restrict__ bias) {
  const int64_t total = M * N;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((N & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)C | (uintptr_t)slabs | (uintptr_t)bias) & 15) == 0) {
    // four columns per thread, eight slabs in flight (same summation order as the scalar form: z ascending per element)
    const int64_t total4 = total >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += stride) {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < splits; z0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int z = z0 + k < splits ? z0 + k : splits - 1;
          v[k] = *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * total + 4 * e);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (z0 + k < splits) sum += v[k];
      }
      const int64_t row = (4 * e) / N, col = 4 * e - row * N;
      f32x4 o = alpha * sum;
      if (bias) o += *reinterpret_cast<const f32x4*>(bias + col);
      f32x4* dst = reinterpret_cast<f32x4*>(C + row * ldc + col);
      if (beta != 0.f) o += beta * *dst;
      *dst = o;
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * total + e];
    const int64_t row = e / N, col = e - row * N;
    float v = alpha * s + (bias ? bias[col] : 0.f);
    if (beta != 0.f) v += beta * C[row * ldc + col];
    C[row * ldc + col] = v;
  }
}

static int pick_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 256 || K <= 4 * BK) return 1;
  int64_t s = 512 / tiles;  // tiles x splits <= 512 = one resident wave of workgroups (2 per CU): no tail round
  const int64_t kt = (K + BK - 1) / BK;
  if (s > kt / 4) s = kt / 4;  // at least 4 k-tiles per split
  if (s > 256) s = 256;
  return s < 1 ? 1 : (int)s;
}

}  // namespace

extern "C" int64_t trs_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = pick_splits(M, N, K);
  return s > 1 ? (int64_t)s * M * N * 4 : 0;
}

static int gemm_impl(bool bf16, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                     const float* A_dev, int64_t lda, const float* B_dev, int64_t ldb, float beta, float* C_dev,
                     int64_t ldc, const float* bias_dev, float* bn_part_dev, void* workspace_dev,
                     int64_t workspace_bytes, void* stream) {
  TRS_REQUIRE(M >= 0 && N >= 0 && K >= 0, "trs_gemm_f32: negative dimension");
  if (M == 0 || N == 0) return TRS_OK;
  TRS_REQUIRE(K > 0, "trs_gemm_f32: K must be positive");
  TRS_REQUIRE(A_dev && B_dev && C_dev, "trs_gemm_f32: NULL operand");
  TRS_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "trs_gemm_f32: leading dimension too small");
  const int splits = bn_part_dev ? 1 : pick_splits(M, N, K);
  TRS_REQUIRE(!bn_part_dev || beta == 0.f, "trs_gemm: fused BatchNorm statistics need beta == 0");
  TRS_REQUIRE(splits == 1 || (workspace_dev && workspace_bytes >= (int64_t)splits * M * N * 4),
              "trs_gemm_f32: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)splits * M * N * 4);
  GemmArgs g;
  g.A = A_dev; g.B = B_dev; g.C = C_dev; g.bias = bias_dev;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  const int64_t kt = (K + BK - 1) / BK;
  g.k_per_split = ((kt + splits - 1) / splits) * BK;
  g.slabs = (float*)workspace_dev;
  g.bn_part = bn_part_dev;
  g.C16 = nullptr;
  g.vecA = (((uintptr_t)A_dev & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  g.vecB = (((uintptr_t)B_dev & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  const int64_t gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
  TRS_REQUIRE(gx * gy * splits < ((int64_t)1 << 31), "trs_gemm_f32: problem too large for the launch grid");
  g.gx = (int)gx; g.gy = (int)gy; g.splits = splits;
  dim3 grid((unsigned)(gx * gy * splits));
  hipStream_t s = (hipStream_t)stream;
  const bool akc = !transA, bkc = transB != 0;
  // all tiles interior: M, N multiples of 128, K a multiple of 32 with at least one k-tile in every split (the last
  // split may be shorter), 16-byte loads legal
  const bool fast = g.vecA && g.vecB && M % BM == 0 && N % BN == 0 && K % BK == 0 &&
                    (int64_t)(splits - 1) * g.k_per_split < K;
  // fp32 NT on the LDS-DMA kernel when 256-row tiles fill the chip (TRS_GEMM32_NO_GLDS=1: tuning / test knob)
  {
    const bool want = trs_tuning().gemm32_no_glds == 0;
    const int bnt = N % 256 == 0 ? 256 : (N % 128 == 0 && !bn_part_dev ? 128 : 0);
    if (!bf16 && want && akc && bkc && splits == 1 && beta == 0.f && bnt && M % 256 == 0 && K % 32 == 0 &&
        (M / 256) * (N / bnt) >= 256 && g.vecA && g.vecB && (((uintptr_t)C_dev) & 15) == 0 && ldc % 4 == 0 &&
        (!bias_dev || true)) {
      g.gx = (int)(N / bnt); g.gy = (int)(M / 256);
      const dim3 grid3((unsigned)((int64_t)g.gx * g.gy));
      const GatherSrc none = {};
      if (bnt == 256) hipLaunchKernelGGL((gemm32_nt_glds_kernel<256, false>), grid3, dim3(512), 0, s, g, none, (float*)nullptr, (int64_t)0);
      else hipLaunchKernelGGL((gemm32_nt_glds_kernel<128, false>), grid3, dim3(512), 0, s, g, none, (float*)nullptr, (int64_t)0);
      TRS_CHECK_LAUNCH("gemm32_nt_glds_kernel");
      return TRS_OK;
    }
  }
  if (bf16) {
#define TRS_GEMM(A_, B_)                                                                         \
  {                                                                                              \
    if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<A_, B_, true>), grid, dim3(256), 0, s, g);    \
    else hipLaunchKernelGGL((gemm_bf16_kernel<A_, B_, false>), grid, dim3(256), 0, s, g);        \
  }
    if (akc && bkc) TRS_GEMM(true, true)
    else if (akc && !bkc) TRS_GEMM(true, false)
    else if (!akc && bkc) TRS_GEMM(false, true)
    else TRS_GEMM(false, false)
#undef TRS_GEMM
  } else {
#define TRS_GEMM(A_, B_)                                                                         \
  {                                                                                              \
    if (fast) hipLaunchKernelGGL((gemm_f32_kernel<A_, B_, true>), grid, dim3(256), 0, s, g);     \
    else hipLaunchKernelGGL((gemm_f32_kernel<A_, B_, false>), grid, dim3(256), 0, s, g);         \
  }
    if (akc && bkc) TRS_GEMM(true, true)
    else if (akc && !bkc) TRS_GEMM(true, false)
    else if (!akc && bkc) TRS_GEMM(false, true)
    else TRS_GEMM(false, false)
#undef TRS_GEMM
  }
  TRS_CHECK_LAUNCH("gemm kernel");
  if (splits > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(trs_grid((M * N + 3) / 4, 256)), dim3(256), 0, s, g.slabs, splits, M, N,
                       C_dev, ldc, alpha, beta, bias_dev);
    TRS_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return TRS_OK;
}

extern "C" int trs_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                            const float* A_dev, int64_t lda, const float* B_dev, int64_t ldb, float beta,
                            float* C_dev, int64_t ldc, const float* bias_dev, float* bn_part_dev,
                            void* workspace_dev, int64_t workspace_bytes, void* stream) {
  return gemm_impl(false, transA, transB, M, N, K, alpha, A_dev, lda, B_dev, ldb, beta, C_dev, ldc, bias_dev,
                   bn_part_dev, workspace_dev, workspace_bytes, stream);
}

extern "C" int trs_gemm_bf16(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                             const float* A_dev, int64_t lda, const float* B_dev, int64_t ldb, float beta,
                             float* C_dev, int64_t ldc, const float* bias_dev, float* bn_part_dev,
                             void* workspace_dev, int64_t workspace_bytes, void* stream) {
  return gemm_impl(true, transA, transB, M, N, K, alpha, A_dev, lda, B_dev, ldb, beta, C_dev, ldc, bias_dev,
                   bn_part_dev, workspace_dev, workspace_bytes, stream);
}

// bf16-resident GEMM.  tn = 0: C(M,N) = alpha * A(M,K) B(N,K)^T; tn = 1: C(M,N) = alpha * A(K,M)^T B(K,N).
// split-K of the weight-gradient (TN) form on 256 x 256 tiles (one workgroup per CU): as many splits as give every CU
// a workgroup, at least 4 k-tiles each
static int tn_wide_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = (M / 256) * (N / 256), kt = K / BK2;
  int64_t s = tiles > 0 ? 256 / tiles : 1;
  if (s > kt / 4) s = kt / 4;
  return s < 1 ? 1 : (int)s;
}
// (measured at the c5 weight-gradient shapes, K = 65 536: 1024 x 1280 281 -> 233 us, 512 x 1024 107 -> 94, but 256 x 512 —
// two tiles x 128 splits of slabs — 38 -> 44: eight tiles at least.  TRS_GEMM16_TN_WIDE = 0 | 1 overrides: tuning knob)
static bool tn_wide_wanted(int64_t M, int64_t N) {
  if (trs_tuning().gemm16_tn_wide >= 0) return trs_tuning().gemm16_tn_wide != 0;
  return (M / 256) * (N / 256) >= 8;
}

extern "C" int64_t trs_gemm_bf16in_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int64_t b = trs_gemm_f32_workspace_bytes(M, N, K);
  if (M > 0 && N > 0 && K > 0 && M % 256 == 0 && N % 256 == 0) {
    const int64_t w = (int64_t)tn_wide_splits(M, N, K) * M * N * 4;
    if (w > b) b = w;
  }
  return b;
}

extern "C" int trs_gemm_bf16in(int tn, int64_t M, int64_t N, int64_t K, float alpha, const void* A_dev, int64_t lda,
                               const void* B_dev, int64_t ldb, float beta, float* C_dev, void* C_bf16_dev, int64_t ldc,
                               const float* bias_dev, float* bn_part_dev, void* workspace_dev,
                               int64_t workspace_bytes, void* stream) {
  TRS_REQUIRE(A_dev && B_dev && (C_dev || C_bf16_dev), "trs_gemm_bf16in: NULL operand");
  TRS_REQUIRE(!C_bf16_dev || (beta == 0.f && !C_dev), "trs_gemm_bf16in: the bf16 output replaces C and needs beta == 0");
  TRS_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK2 == 0,
              "trs_gemm_bf16in: needs M, N multiples of 128 and K a multiple of 64 (got %lld x %lld x %lld)",
              (long long)M, (long long)N, (long long)K);
  TRS_REQUIRE(lda >= (tn ? M : K) && ldb >= (tn ? N : K) && ldc >= N, "trs_gemm_bf16in: leading dimension too small");
  TRS_REQUIRE((((uintptr_t)A_dev | (uintptr_t)B_dev) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0,
              "trs_gemm_bf16in: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  int splits = (bn_part_dev || C_bf16_dev) ? 1 : pick_splits(M, N, K);
  const bool tn_wide = tn && !bn_part_dev && !C_bf16_dev && M % 256 == 0 && N % 256 == 0 && tn_wide_wanted(M, N) &&
                       (M / 256) * (N / 256) * tn_wide_splits(M, N, K) >= 128;
  if (tn_wide) splits = tn_wide_splits(M, N, K);
  const int64_t kt = K / BK2;
  if (splits > kt) splits = (int)kt;
  int64_t per = (kt + splits - 1) / splits;
  splits = (int)((kt + per - 1) / per);  // every split gets at least one k-tile
  TRS_REQUIRE(!bn_part_dev || beta == 0.f, "trs_gemm_bf16in: fused BatchNorm statistics need beta == 0");
  TRS_REQUIRE(splits == 1 || (workspace_dev && workspace_bytes >= (int64_t)splits * M * N * 4),
              "trs_gemm_bf16in: workspace too small");
  Gemm16Args g;
  g.A = (const unsigned short*)A_dev; g.B = (const unsigned short*)B_dev; g.C = C_dev; g.bias = bias_dev;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
  g.k_per_split = per * BK2;
  g.slabs = (float*)workspace_dev;
  g.bn_part = bn_part_dev;
  g.C16 = (unsigned short*)C_bf16_dev;
  // 256-row tiles when they still give every CU a workgroup (TRS_GEMM16_BM = 128 | 256 overrides: tuning knob)
  // Tile selection (TRS_GEMM16_TILE = 128 | 256 | 512 forces 128x128 | 256x128 | 256x256 where the shape allows: tests,
  // tuning).  The kernel is bound by the vector L1's miss path (TCP_PENDING_STALL ~ 50 % of the cycles at 128x128), so
  // the tile that moves the fewest operand bytes per MAC wins as long as every CU still gets a workgroup.
  const int tile = trs_tuning().gemm16_tile;
  const bool can_big = M % 256 == 0, can_wide = can_big && N % 256 == 0;
  const bool wide = tn_wide || (tile ? (tile == 512 && can_wide) : (can_wide && (M / 256) * (N / 256) * splits >= 256));
  const bool big = wide || (tile ? (tile == 256 && can_big) : (can_big && (M / 256) * (N / BN) * splits >= 256));
  const int64_t gx = N / (wide ? 256 : BN), gy = M / (big ? 256 : BM);
  TRS_REQUIRE(gx * gy * splits < ((int64_t)1 << 31), "trs_gemm_bf16in: problem too large for the launch grid");
  g.gx = (int)gx; g.gy = (int)gy; g.splits = splits;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(gx * gy * splits));
  const bool no_glds = trs_tuning().gemm16_no_glds != 0;  // A/B knob (tests, tuning): the register-staged 256 x 256 kernel
  const bool c_vec = (((uintptr_t)(C_bf16_dev ? C_bf16_dev : (void*)C_dev)) & 15) == 0 && ldc % 8 == 0;  // 16-byte stores
  if (wide && !tn && splits == 1 && beta == 0.f && !no_glds && c_vec) {
    g.gx = (int)(N / 256); g.gy = (int)(M / 256);
    const dim3 grid3((unsigned)((int64_t)g.gx * g.gy));
    if (g.C16) hipLaunchKernelGGL((gemm16_nt_glds_kernel<true>), grid3, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm16_nt_glds_kernel<false>), grid3, dim3(512), 0, s, g);
    TRS_CHECK_LAUNCH("gemm16_nt_glds_kernel");
    return TRS_OK;
  }
  if (wide && tn) hipLaunchKernelGGL((gemm_bf16in_kernel<true, 256, 256>), grid, dim3(1024), 0, s, g);
  else if (wide) hipLaunchKernelGGL((gemm_bf16in_kernel<false, 256, 256>), grid, dim3(1024), 0, s, g);
  else if (tn && big) hipLaunchKernelGGL((gemm_bf16in_kernel<true, 256>), grid, dim3(512), 0, s, g);
  else if (tn) hipLaunchKernelGGL((gemm_bf16in_kernel<true, 128>), grid, dim3(256), 0, s, g);
  else if (big) hipLaunchKernelGGL((gemm_bf16in_kernel<false, 256>), grid, dim3(512), 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16in_kernel<false, 128>), grid, dim3(256), 0, s, g);
  TRS_CHECK_LAUNCH("gemm_bf16in_kernel");
  if (splits > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(trs_grid((M * N + 3) / 4, 256)), dim3(256), 0, s, g.slabs, splits, M, N,
                       C_dev, ldc, alpha, beta, bias_dev);
    TRS_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return TRS_OK;
}

// MLP layer 0 forward with the gather fused into the GEMM (include/trs.h).  Returns 1 (not an error) when the shape is
// not one the fused kernels take: the caller then runs trs_mlp_gather_concat + trs_gemm_*.
extern "C" int trs_mlp_gather_gemm1_fwd(const trs_tables* tables, const trs_batch* batch, int32_t passes, int32_t bf16,
                                        const void* W_dev, int64_t ldw, const float* bias_dev, int64_t N, float* y_dev,
                                        void* y_bf16_dev, int64_t ldy, float* bn_part_dev, float* x_dev,
                                        void* x_bf16_dev, int64_t ldx, void* stream) {
  TRS_REQUIRE(tables && batch && W_dev && (y_dev || y_bf16_dev), "trs_mlp_gather_gemm1_fwd: NULL argument");
  TRS_REQUIRE(passes == 1 || passes == 2, "trs_mlp_gather_gemm1_fwd: passes must be 1 or 2");
  TRS_REQUIRE(tables->M >= 0 && tables->M <= TRS_MAX_META, "trs_mlp_gather_gemm1_fwd: bad M");
  const int64_t B = batch->B, rows = (int64_t)passes * B;
  const int D = tables->D, F = 2 + tables->M;
  const int64_t K = (int64_t)F * D;
  const int tile_k = bf16 ? BK2 : 32;
  if (batch->idx_bytes != 4 || B <= 0 || B % 256 != 0 || D % tile_k != 0 || N % 256 != 0 ||
      (rows / 256) * (N / 256) < 256 || !tables->user || !tables->item || (passes == 2 && !batch->neg) ||
      (tables->M > 0 && (!batch->pos_meta || (passes == 2 && !batch->neg_meta))))
    return 1;
  if (bf16 ? (!y_bf16_dev && !y_dev) || (y_bf16_dev && y_dev) || x_dev : (!y_dev || y_bf16_dev || x_bf16_dev)) return 1;
  const void* yp = bf16 && y_bf16_dev ? y_bf16_dev : (void*)y_dev;
  if ((((uintptr_t)W_dev | (uintptr_t)yp | (uintptr_t)x_dev | (uintptr_t)x_bf16_dev) & 15) != 0 || ldw < K ||
      ldw % 8 != 0 || ldy < N || ldy % 8 != 0 || ((x_dev || x_bf16_dev) && (ldx < K || ldx % 8 != 0)))
    return 1;
  GatherSrc S = {};
  S.B = B; S.D = D; S.F = F; S.err = batch->err_flag_dev;
  S.tab[0] = tables->user; S.n_rows[0] = tables->n_users; S.stride[0] = 1;
  S.tab[1] = tables->item; S.n_rows[1] = tables->n_items; S.stride[1] = 1;
  S.ids[0][0] = S.ids[1][0] = (const int32_t*)batch->user;
  S.ids[0][1] = (const int32_t*)batch->pos;
  S.ids[1][1] = (const int32_t*)batch->neg;
  for (int m = 0; m < tables->M; ++m) {
    if (!tables->meta[m] || ((uintptr_t)tables->meta[m] & 15) != 0) return 1;
    S.tab[2 + m] = tables->meta[m]; S.n_rows[2 + m] = tables->n_meta[m]; S.stride[2 + m] = tables->M;
    S.ids[0][2 + m] = (const int32_t*)batch->pos_meta + m;
    S.ids[1][2 + m] = batch->neg_meta ? (const int32_t*)batch->neg_meta + m : nullptr;
  }
  if ((((uintptr_t)tables->user | (uintptr_t)tables->item) & 15) != 0) return 1;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows / 256) * (N / 256)));
  if (bf16) {
    Gemm16Args g = {};
    g.B = (const unsigned short*)W_dev; g.C = y_dev; g.bias = bias_dev;
    g.M = rows; g.N = N; g.K = K; g.ldb = ldw; g.ldc = ldy; g.alpha = 1.f; g.beta = 0.f;
    g.k_per_split = K; g.gx = (int)(N / 256); g.gy = (int)(rows / 256); g.splits = 1;
    g.bn_part = bn_part_dev; g.C16 = (unsigned short*)y_bf16_dev;
    if (g.C16) hipLaunchKernelGGL((gemm16_gather_nt_kernel<true>), grid, dim3(512), 0, s, g, S, (unsigned short*)x_bf16_dev, ldx);
    else hipLaunchKernelGGL((gemm16_gather_nt_kernel<false>), grid, dim3(512), 0, s, g, S, (unsigned short*)x_bf16_dev, ldx);
    TRS_CHECK_LAUNCH("gemm16_gather_nt_kernel");
  } else {
    GemmArgs g = {};
    g.B = (const float*)W_dev; g.C = y_dev; g.bias = bias_dev;
    g.M = rows; g.N = N; g.K = K; g.ldb = ldw; g.ldc = ldy; g.alpha = 1.f; g.beta = 0.f;
    g.k_per_split = K; g.gx = (int)(N / 256); g.gy = (int)(rows / 256); g.splits = 1; g.vecA = g.vecB = 1;
    g.bn_part = bn_part_dev;
    hipLaunchKernelGGL((gemm32_nt_glds_kernel<256, true>), grid, dim3(512), 0, s, g, S, x_dev, ldx);
    TRS_CHECK_LAUNCH("gemm32_nt_glds_kernel<gather>");
  }
  return TRS_OK;
}

extern "C" int trs_f32_to_bf16(const float* src_dev, int64_t rows, int64_t cols, int64_t ld, void* dst_dev,
                               void* dst_t_dev, void* stream) {
  TRS_REQUIRE(src_dev && (dst_dev || dst_t_dev) && rows > 0 && cols > 0 && ld >= cols, "trs_f32_to_bf16: bad arguments");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(f32_to_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, src_dev, rows, cols, ld,
                     (unsigned short*)dst_dev, (unsigned short*)dst_t_dev);
  TRS_CHECK_LAUNCH("f32_to_bf16_kernel");
  return TRS_OK;
}

extern "C" int trs_f32_to_bf16_multi(int32_t n, const float* const* src_dev, const int64_t* rows, const int64_t* cols,
                                     const int64_t* ld, void* const* dst_dev, void* const* dst_t_dev, void* stream) {
  TRS_REQUIRE(n >= 1 && n <= TRS_IMG_MAX && src_dev && rows && cols && ld && dst_dev && dst_t_dev,
              "trs_f32_to_bf16_multi: 1..%d matrices", TRS_IMG_MAX);
  ImgArgs a = {};
  a.n = n;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    TRS_REQUIRE(src_dev[k] && (dst_dev[k] || dst_t_dev[k]) && rows[k] > 0 && cols[k] > 0 && ld[k] >= cols[k],
                "trs_f32_to_bf16_multi: bad arguments for matrix %d", k);
    a.src[k] = src_dev[k]; a.dst[k] = (unsigned short*)dst_dev[k]; a.dst_t[k] = (unsigned short*)dst_t_dev[k];
    a.rows[k] = rows[k]; a.cols[k] = cols[k]; a.ld[k] = ld[k];
    a.first[k] = total;
    total += ((rows[k] + 31) / 32) * ((cols[k] + 31) / 32);
  }
  a.first[n] = total;
  TRS_REQUIRE(total < ((int64_t)1 << 31), "trs_f32_to_bf16_multi: too many tiles");
  hipLaunchKernelGGL(f32_to_bf16_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  TRS_CHECK_LAUNCH("f32_to_bf16_multi_kernel");
  return TRS_OK;
}
