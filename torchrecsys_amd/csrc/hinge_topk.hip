// hinge_topk.hip — hinge loss / pairwise AUC reductions (reference helper/loss.py:5-9, evaluate/metrics.py:23-31)
// and the predict() top-k (reference model.py:447-450: torch.sort(descending=True)[:top_k]).
#include "trs_common.h"

namespace {

__global__ __launch_bounds__(TRS_BLOCK) void hinge_auc_kernel(const float* __restrict__ pos,
                                                             const float* __restrict__ neg, int64_t B,
                                                             float* loss_sum, int32_t* auc_count, int loss,
                                                             float inv_B, float* __restrict__ gpos,
                                                             float* __restrict__ gneg) {
  float L = 0.f;
  int A = 0;
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t t = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; t < B; t += stride) {
    const float p = pos[t], n = neg[t];
    float lval, dneg;
    trs_pair_loss(loss, p, n, lval, dneg);
    L += lval;
    A += (p > n) ? 1 : 0;
    if (gpos) {  // trs_hinge_auc_backward: the gradient of the mean loss in the same sweep
      const float act = dneg * inv_B;
      gpos[t] = -act;
      gneg[t] = act;
    }
  }
  __shared__ float s_l[TRS_BLOCK / TRS_WAVE];
  __shared__ int s_a[TRS_BLOCK / TRS_WAVE];
  L = trs_wave_sum(L);
  A = trs_wave_sum_i(A);
  if ((threadIdx.x & 63) == 0) {
    s_l[threadIdx.x >> 6] = L;
    s_a[threadIdx.x >> 6] = A;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    int a = 0;
    for (int w = 0; w < TRS_BLOCK / TRS_WAVE; ++w) {
      l += s_l[w];
      a += s_a[w];
    }
    if (loss_sum && l != 0.f) atomicAdd(loss_sum, l);
    if (auc_count && a != 0) atomicAdd(auc_count, a);
  }
}

// The same reductions for consecutive batches of `batch` rows in one launch (evaluate(): blockIdx.y = batch).
__global__ __launch_bounds__(TRS_BLOCK) void hinge_auc_batches_kernel(const float* __restrict__ pos,
                                                                     const float* __restrict__ neg, int64_t n_total,
                                                                     int64_t batch, float* loss_sums,
                                                                     int32_t* auc_counts, int loss) {
  const int64_t b = blockIdx.y;
  const int64_t t0 = b * batch, t1 = t0 + batch < n_total ? t0 + batch : n_total;
  float L = 0.f;
  int A = 0;
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t t = t0 + (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; t < t1; t += stride) {
    const float p = pos[t], n = neg[t];
    float lval, dneg;
    trs_pair_loss(loss, p, n, lval, dneg);
    L += lval;
    A += (p > n) ? 1 : 0;
  }
  __shared__ float s_l[TRS_BLOCK / TRS_WAVE];
  __shared__ int s_a[TRS_BLOCK / TRS_WAVE];
  L = trs_wave_sum(L);
  A = trs_wave_sum_i(A);
  if ((threadIdx.x & 63) == 0) {
    s_l[threadIdx.x >> 6] = L;
    s_a[threadIdx.x >> 6] = A;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    int a = 0;
    for (int w = 0; w < TRS_BLOCK / TRS_WAVE; ++w) {
      l += s_l[w];
      a += s_a[w];
    }
    if (loss_sums && l != 0.f) atomicAdd(loss_sums + b, l);
    if (auc_counts && a != 0) atomicAdd(auc_counts + b, a);
  }
}

__global__ __launch_bounds__(TRS_BLOCK) void hinge_backward_kernel(const float* __restrict__ pos,
                                                                  const float* __restrict__ neg, int64_t B,
                                                                  float inv_B, float* __restrict__ gpos,
                                                                  float* __restrict__ gneg, int loss) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t t = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; t < B; t += stride) {
    float lval, dneg;
    trs_pair_loss(loss, pos[t], neg[t], lval, dneg);
    const float act = dneg * inv_B;
    gpos[t] = -act;
    gneg[t] = act;
  }
}

// ------------------------------------------------------------------------------------------------ top-k
// key = (orderable(score) << 32) | (0xFFFFFFFF - index): descending key order == (score desc, index asc).
// NaN sorts first (torch.sort(descending=True) treats NaN as the largest value); -0.0 ties with +0.0.
constexpr int TOPK_CHUNK = 4096;
constexpr int TOPK_MAXK = TOPK_CHUNK / 2;

__device__ __forceinline__ uint64_t topk_key(float s, uint32_t idx) {
  uint32_t o;
  if (s != s) {
    o = 0xFFFFFFFFu;
  } else {
    if (s == 0.f) s = 0.f;
    const uint32_t b = __float_as_uint(s);
    o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
  return ((uint64_t)o << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}

// Each block sorts one chunk of TOPK_CHUNK keys (descending, bitonic in LDS) and emits its first k.
// FROM_SCORES: read fp32 scores and build keys; otherwise read keys.  FINAL: write int64 indices instead of keys.
template <bool FROM_SCORES, bool FINAL>
__global__ __launch_bounds__(TRS_BLOCK) void topk_chunk_kernel(const float* __restrict__ scores,
                                                              const uint64_t* __restrict__ keys_in, int64_t n, int k,
                                                              uint64_t* __restrict__ keys_out,
                                                              int64_t* __restrict__ idx_out) {
  __shared__ uint64_t s[TOPK_CHUNK];
  const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
  for (int i = threadIdx.x; i < TOPK_CHUNK; i += TRS_BLOCK) {
    const int64_t g = base + i;
    uint64_t key = 0;  // below every real key
    if (g < n) key = FROM_SCORES ? topk_key(scores[g], (uint32_t)g) : keys_in[g];
    s[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= TOPK_CHUNK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < TOPK_CHUNK / 2; i += TRS_BLOCK) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;  // descending blocks first -> whole array descending at the end
        const uint64_t a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += TRS_BLOCK) {
    if (FINAL)
      idx_out[i] = (int64_t)(0xFFFFFFFFu - (uint32_t)(s[i] & 0xFFFFFFFFu));
    else
      keys_out[(int64_t)blockIdx.x * k + i] = s[i];
  }
}

// ---- k > TOPK_MAXK (predict(top_k) up to n_items, reference model.py:447-450 sorts everything): a full bitonic sort
// of the P = 2^ceil(log2 n) padded keys in global memory, descending.  Strides >= TOPK_CHUNK are one pass over the
// keys each (sort_global_step_kernel); all strides below TOPK_CHUNK of one merge size run in LDS (sort_local_kernel,
// which also performs every merge size up to TOPK_CHUNK in its first call).  Off the hot path: ~45 launches at 1M items.
__global__ __launch_bounds__(TRS_BLOCK) void sort_keys_kernel(const float* __restrict__ scores, int64_t n, int64_t P,
                                                             uint64_t* __restrict__ keys) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t g = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; g < P; g += stride)
    keys[g] = g < n ? topk_key(scores[g], (uint32_t)g) : 0;  // padding sorts below every real key
}

__global__ __launch_bounds__(TRS_BLOCK) void sort_global_step_kernel(uint64_t* __restrict__ keys, int64_t P,
                                                                    int64_t size, int64_t stride) {
  const int64_t gs = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; i < P / 2; i += gs) {
    const int64_t lo = 2 * i - (i & (stride - 1));
    const int64_t hi = lo + stride;
    const bool desc = (lo & size) == 0;
    const uint64_t a = keys[lo], b = keys[hi];
    if ((a < b) == desc) {
      keys[lo] = b;
      keys[hi] = a;
    }
  }
}

// One TOPK_CHUNK-aligned chunk per block: merge sizes first_size .. last_size (last_size > TOPK_CHUNK: only that size,
// strides < TOPK_CHUNK — the wider strides were done by the global steps).
__global__ __launch_bounds__(TRS_BLOCK) void sort_local_kernel(uint64_t* __restrict__ keys, int64_t first_size,
                                                              int64_t last_size) {
  __shared__ uint64_t s[TOPK_CHUNK];
  const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
  for (int i = threadIdx.x; i < TOPK_CHUNK; i += TRS_BLOCK) s[i] = keys[base + i];
  __syncthreads();
  for (int64_t size = first_size; size <= last_size; size <<= 1) {
    for (int stride = (int)(size / 2 < TOPK_CHUNK / 2 ? size / 2 : TOPK_CHUNK / 2); stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < TOPK_CHUNK / 2; i += TRS_BLOCK) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((base + lo) & size) == 0;
        const uint64_t a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < TOPK_CHUNK; i += TRS_BLOCK) keys[base + i] = s[i];
}

__global__ __launch_bounds__(TRS_BLOCK) void sort_emit_kernel(const uint64_t* __restrict__ keys, int64_t k,
                                                             int64_t* __restrict__ idx_out) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; i < k; i += stride)
    idx_out[i] = (int64_t)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu));
}

static int64_t pow2_at_least(int64_t n, int64_t lo) {
  int64_t p = lo;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace

extern "C" int trs_hinge_auc(const float* pos_dev, const float* neg_dev, int64_t B, float* loss_sum_dev,
                             int32_t* auc_count_dev, int32_t loss, void* stream) {
  TRS_REQUIRE(loss == TRS_LOSS_HINGE || loss == TRS_LOSS_BPR, "trs_hinge_auc: bad loss kind");
  TRS_REQUIRE(B >= 0, "trs_hinge_auc: negative B");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(pos_dev && neg_dev, "trs_hinge_auc: scores are NULL");
  hipLaunchKernelGGL(hinge_auc_kernel, dim3(trs_grid(B, TRS_BLOCK * 4)), dim3(TRS_BLOCK), 0, (hipStream_t)stream,
                     pos_dev, neg_dev, B, loss_sum_dev, auc_count_dev, (int)loss, 0.f, (float*)nullptr, (float*)nullptr);
  TRS_CHECK_LAUNCH("hinge_auc_kernel");
  return TRS_OK;
}

extern "C" int trs_hinge_auc_backward(const float* pos_dev, const float* neg_dev, int64_t B, float inv_B,
                                      float* loss_sum_dev, int32_t* auc_count_dev, float* gpos_dev, float* gneg_dev,
                                      int32_t loss, void* stream) {
  TRS_REQUIRE(loss == TRS_LOSS_HINGE || loss == TRS_LOSS_BPR, "trs_hinge_auc_backward: bad loss kind");
  TRS_REQUIRE(B >= 0, "trs_hinge_auc_backward: negative B");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(pos_dev && neg_dev && gpos_dev && gneg_dev, "trs_hinge_auc_backward: NULL argument");
  hipLaunchKernelGGL(hinge_auc_kernel, dim3(trs_grid(B, TRS_BLOCK * 4)), dim3(TRS_BLOCK), 0, (hipStream_t)stream,
                     pos_dev, neg_dev, B, loss_sum_dev, auc_count_dev, (int)loss, inv_B, gpos_dev, gneg_dev);
  TRS_CHECK_LAUNCH("hinge_auc_kernel");
  return TRS_OK;
}

extern "C" int trs_hinge_auc_batches(const float* pos_dev, const float* neg_dev, int64_t n_total, int64_t batch,
                                     float* loss_sums_dev, int32_t* auc_counts_dev, int32_t loss, void* stream) {
  TRS_REQUIRE(loss == TRS_LOSS_HINGE || loss == TRS_LOSS_BPR, "trs_hinge_auc_batches: bad loss kind");
  TRS_REQUIRE(n_total >= 0 && batch > 0, "trs_hinge_auc_batches: bad sizes");
  if (n_total == 0) return TRS_OK;
  TRS_REQUIRE(pos_dev && neg_dev, "trs_hinge_auc_batches: scores are NULL");
  const int64_t nb = (n_total + batch - 1) / batch;
  TRS_REQUIRE(nb <= 65535, "trs_hinge_auc_batches: more than 65535 batches in one call");
  hipLaunchKernelGGL(hinge_auc_batches_kernel, dim3(trs_grid(batch, TRS_BLOCK * 4), (unsigned)nb), dim3(TRS_BLOCK), 0,
                     (hipStream_t)stream, pos_dev, neg_dev, n_total, batch, loss_sums_dev, auc_counts_dev, (int)loss);
  TRS_CHECK_LAUNCH("hinge_auc_batches_kernel");
  return TRS_OK;
}

extern "C" int trs_hinge_backward(const float* pos_dev, const float* neg_dev, int64_t B, float inv_B,
                                  float* gpos_dev, float* gneg_dev, int32_t loss, void* stream) {
  TRS_REQUIRE(loss == TRS_LOSS_HINGE || loss == TRS_LOSS_BPR, "trs_hinge_backward: bad loss kind");
  TRS_REQUIRE(B >= 0, "trs_hinge_backward: negative B");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(pos_dev && neg_dev && gpos_dev && gneg_dev, "trs_hinge_backward: NULL argument");
  hipLaunchKernelGGL(hinge_backward_kernel, dim3(trs_grid(B, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream,
                     pos_dev, neg_dev, B, inv_B, gpos_dev, gneg_dev, (int)loss);
  TRS_CHECK_LAUNCH("hinge_backward_kernel");
  return TRS_OK;
}

extern "C" int64_t trs_topk_workspace_bytes(int64_t n, int32_t k) {
  if (n <= 0 || k <= 0) return 0;
  if (k > TOPK_MAXK) return 8 * pow2_at_least(n, TOPK_CHUNK);  // full sort of the padded keys
  const int64_t nb0 = (n + TOPK_CHUNK - 1) / TOPK_CHUNK;
  const int64_t nb1 = (nb0 * k + TOPK_CHUNK - 1) / TOPK_CHUNK;
  return 8 * (int64_t)k * (nb0 + nb1);
}

extern "C" int trs_topk(const float* scores_dev, int64_t n, int32_t k, int64_t* idx_out_dev, void* workspace_dev,
                        int64_t workspace_bytes, void* stream) {
  TRS_REQUIRE(n > 0 && n < ((int64_t)1 << 32), "trs_topk: n=%lld outside 1..2^32-1", (long long)n);
  TRS_REQUIRE(k >= 1 && k <= n, "trs_topk: need 1 <= k <= n (k=%d, n=%lld)", k, (long long)n);
  TRS_REQUIRE(scores_dev && idx_out_dev, "trs_topk: scores/idx_out is NULL");
  TRS_REQUIRE(workspace_bytes >= trs_topk_workspace_bytes(n, k) && (workspace_dev || (n <= TOPK_CHUNK && k <= TOPK_MAXK)),
              "trs_topk: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)trs_topk_workspace_bytes(n, k));
  hipStream_t s = (hipStream_t)stream;
  if (k > TOPK_MAXK) {  // more than a chunk can hand on: sort everything (the reference's torch.sort, model.py:447)
    const int64_t P = pow2_at_least(n, TOPK_CHUNK);
    uint64_t* keys = (uint64_t*)workspace_dev;
    const dim3 bl(TRS_BLOCK), gp(trs_grid(P, TRS_BLOCK)), gh(trs_grid(P / 2, TRS_BLOCK)), gc((unsigned)(P / TOPK_CHUNK));
    hipLaunchKernelGGL(sort_keys_kernel, gp, bl, 0, s, scores_dev, n, P, keys);
    hipLaunchKernelGGL(sort_local_kernel, gc, bl, 0, s, keys, (int64_t)2, (int64_t)TOPK_CHUNK);
    for (int64_t size = 2 * TOPK_CHUNK; size <= P; size <<= 1) {
      for (int64_t stride = size / 2; stride >= TOPK_CHUNK; stride >>= 1)
        hipLaunchKernelGGL(sort_global_step_kernel, gh, bl, 0, s, keys, P, size, stride);
      hipLaunchKernelGGL(sort_local_kernel, gc, bl, 0, s, keys, size, size);
    }
    hipLaunchKernelGGL(sort_emit_kernel, dim3(trs_grid(k, TRS_BLOCK)), bl, 0, s, keys, (int64_t)k, idx_out_dev);
    TRS_CHECK_LAUNCH("topk full sort");
    return TRS_OK;
  }
  if (n <= TOPK_CHUNK) {
    hipLaunchKernelGGL((topk_chunk_kernel<true, true>), dim3(1), dim3(TRS_BLOCK), 0, s, scores_dev, nullptr, n, k,
                       nullptr, idx_out_dev);
    TRS_CHECK_LAUNCH("topk_chunk_kernel");
    return TRS_OK;
  }
  const int64_t nb0 = (n + TOPK_CHUNK - 1) / TOPK_CHUNK;
  uint64_t* bufA = (uint64_t*)workspace_dev;
  uint64_t* bufB = bufA + nb0 * k;
  hipLaunchKernelGGL((topk_chunk_kernel<true, false>), dim3((unsigned)nb0), dim3(TRS_BLOCK), 0, s, scores_dev,
                     nullptr, n, k, bufA, nullptr);
  TRS_CHECK_LAUNCH("topk_chunk_kernel");
  int64_t cnt = nb0 * k;
  uint64_t *in = bufA, *out = bufB;
  while (cnt > TOPK_CHUNK) {
    const int64_t nb = (cnt + TOPK_CHUNK - 1) / TOPK_CHUNK;
    hipLaunchKernelGGL((topk_chunk_kernel<false, false>), dim3((unsigned)nb), dim3(TRS_BLOCK), 0, s, nullptr, in,
                       cnt, k, out, nullptr);
    TRS_CHECK_LAUNCH("topk_chunk_kernel");
    cnt = nb * k;
    uint64_t* tmp = in;
    in = out;
    out = tmp;
  }
  hipLaunchKernelGGL((topk_chunk_kernel<false, true>), dim3(1), dim3(TRS_BLOCK), 0, s, nullptr, in, cnt, k, nullptr,
                     idx_out_dev);
  TRS_CHECK_LAUNCH("topk_chunk_kernel");
  return TRS_OK;
}
