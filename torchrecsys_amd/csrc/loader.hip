// loader.hip — device side of FastDataLoader (reference dataset/dataset.py:319-458): epoch shuffle, batch slice,
// dynamic negative sampler.  Integer work: results are bit-exact against oracle/loader.py.
#include "trs_common.h"

namespace {

__global__ __launch_bounds__(TRS_BLOCK) void sample_neg_kernel(const void* __restrict__ pos, int ib, int64_t B,
                                                              int64_t n_items, uint64_t seed, uint64_t offset,
                                                              void* __restrict__ neg) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t t = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; t < B; t += stride) {
    const int64_t p = trs_ld_idx(pos, ib, t);
    trs_st_idx(neg, ib, t, trs_sample_one_neg(seed, offset + (uint64_t)t, p, n_items));
  }
}

struct PrepArgs {
  const int32_t* su;
  const int32_t* si;
  const int32_t* neg_static;
  int64_t N;
  uint64_t shuffle_key;
  int hb;
  int64_t t0, B, n_items;
  uint64_t seed, offset;
  const int32_t* item_meta;
  int M;
  int32_t *user, *pos, *neg, *pos_meta, *neg_meta;
  TrsSampler S;
};

__global__ __launch_bounds__(TRS_BLOCK) void batch_prepare_kernel(const PrepArgs a) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t t = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; t < a.B; t += stride) {
    const int64_t q = a.t0 + t;
    const int64_t p = trs_feistel_perm(q, a.N * a.S.k_neg, a.shuffle_key, a.hb) % a.N;
    const int32_t u = a.su[p];
    const int32_t i = a.si[p];
    int32_t j;
    if (a.neg_static)
      j = a.neg_static[p];
    else if (a.S.max_tries != 0 && (uint64_t)i >= (uint64_t)a.n_items)
      j = 0;  // (reported by the scorer; the option paths index tables by the ids)
    else
      j = (int32_t)trs_sample_neg_opt(a.seed, a.offset + (uint64_t)t, (int64_t)u, (int64_t)i, a.n_items, a.S);
    a.user[t] = u;
    a.pos[t] = i;
    a.neg[t] = j;
    if (a.M > 0) {
      // an id outside the item table is reported by the scorer kernel; keep this lookup in range
      const int64_t ic = ((uint64_t)i < (uint64_t)a.n_items) ? i : 0;
      const int64_t jc = ((uint64_t)j < (uint64_t)a.n_items) ? j : 0;
      for (int m = 0; m < a.M; ++m) {
        a.pos_meta[t * a.M + m] = a.item_meta[ic * a.M + m];
        a.neg_meta[t * a.M + m] = a.item_meta[jc * a.M + m];
      }
    }
  }
}

}  // namespace

extern "C" int trs_sample_neg(const void* pos_dev, int idx_bytes, int64_t B, int64_t n_items, uint64_t seed,
                              uint64_t offset, void* neg_out_dev, void* stream) {
  TRS_REQUIRE(idx_bytes == 4 || idx_bytes == 8, "trs_sample_neg: idx_bytes must be 4 or 8");
  TRS_REQUIRE(B >= 0, "trs_sample_neg: negative B");
  TRS_REQUIRE(n_items >= 2, "trs_sample_neg: need n_items >= 2 to draw a negative different from the positive");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(pos_dev && neg_out_dev, "trs_sample_neg: pos/neg is NULL");
  hipLaunchKernelGGL(sample_neg_kernel, dim3(trs_grid(B, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream,
                     pos_dev, idx_bytes, B, n_items, seed, offset, neg_out_dev);
  TRS_CHECK_LAUNCH("sample_neg_kernel");
  return TRS_OK;
}

extern "C" int trs_batch_prepare(const int32_t* stream_user_dev, const int32_t* stream_item_dev,
                                 const int32_t* neg_static_dev, int64_t N, uint64_t shuffle_key, int64_t t0,
                                 int64_t B, int64_t n_items, uint64_t sample_seed, uint64_t sample_offset,
                                 const int32_t* item_meta_dev, int32_t M, int32_t* user_out, int32_t* pos_out,
                                 int32_t* neg_out, int32_t* pos_meta_out, int32_t* neg_meta_out,
                                 const trs_sampler* sampler, void* stream) {
  const int64_t kn = sampler && sampler->k_neg > 1 ? sampler->k_neg : 1;
  TRS_REQUIRE(N > 0 && t0 >= 0 && B >= 0 && t0 + B <= N * kn, "trs_batch_prepare: slice [%lld,%lld) outside [0,%lld)",
              (long long)t0, (long long)(t0 + B), (long long)(N * kn));
  TRS_REQUIRE(!sampler || (sampler->k_neg >= 1 && (!sampler->popularity || (sampler->pop_items && sampler->pop_n > 0)) &&
                           ((sampler->seen_off == nullptr) == (sampler->seen_items == nullptr))),
              "trs_batch_prepare: bad sampler options");
  TRS_REQUIRE(M >= 0 && M <= TRS_MAX_META, "trs_batch_prepare: bad M");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(stream_user_dev && stream_item_dev, "trs_batch_prepare: stream is NULL");
  TRS_REQUIRE(user_out && pos_out && neg_out, "trs_batch_prepare: outputs are NULL");
  TRS_REQUIRE(neg_static_dev || n_items >= 2, "trs_batch_prepare: dynamic sampling needs n_items >= 2");
  TRS_REQUIRE(M == 0 || (item_meta_dev && pos_meta_out && neg_meta_out),
              "trs_batch_prepare: M=%d needs item_meta and metadata outputs", M);
  PrepArgs a = {stream_user_dev, stream_item_dev, neg_static_dev, N, shuffle_key, trs_feistel_half_bits(N * kn), t0, B,
                n_items, sample_seed, sample_offset, item_meta_dev, M, user_out, pos_out, neg_out, pos_meta_out,
                neg_meta_out, trs_sampler_args(sampler)};
  hipLaunchKernelGGL(batch_prepare_kernel, dim3(trs_grid(B, TRS_BLOCK)), dim3(TRS_BLOCK), 0, (hipStream_t)stream, a);
  TRS_CHECK_LAUNCH("batch_prepare_kernel");
  return TRS_OK;
}
