// rows.hip — sparse embedding-row optimiser kernels.
//
// The reference hands autograd's uncoalesced sparse COO gradients to torch.optim (model.py:188-200).  Here the staged
// per-triple gradient rows (score_kernels.h MODE 1, or the MLP's d x0) ARE those COO values; these kernels apply them.
//
// Atomic shape (MI355X_MICROARCH.md "Global float atomics"): one global_atomic_add_f32 wave-instruction covers
// 256 contiguous bytes of one row (or 2 x 128 B of two rows) — never one lane per row.
#include "trs_common.h"

namespace {

// lanes-per-row LPR (power of two, <= 64); a wave handles 64/LPR entries per iteration.
template <int LPR>
__global__ __launch_bounds__(TRS_BLOCK) void rows_scatter_add_kernel(float* __restrict__ table, int64_t n_rows, int D,
                                                                    const void* __restrict__ idx, int ib,
                                                                    const float* __restrict__ vals, int64_t ld,
                                                                    int64_t n, float alpha, int32_t* err) {
  constexpr int EPW = TRS_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int lir = lane % LPR;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t niter = (n + EPW - 1) / EPW;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t e = it * EPW + lane / LPR;
    if (e >= n) continue;
    const int64_t row = trs_ld_idx(idx, ib, e);
    if ((uint64_t)row >= (uint64_t)n_rows) {
      if (lir == 0 && err) atomicOr(err, 1);
      continue;
    }
    float* dst = table + row * (int64_t)D;
    const float* src = vals + e * ld;
    for (int d = lir; d < D; d += LPR) atomicAdd(dst + d, alpha * src[d]);
  }
}

template <int LPR>
static void launch_scatter(float* table, int64_t n_rows, int D, const void* idx, int ib, const float* vals,
                           int64_t ld, int64_t n, float alpha, int32_t* err, hipStream_t s) {
  constexpr int EPW = TRS_WAVE / LPR;
  const int64_t waves = (n + EPW - 1) / EPW;
  const int grid = trs_grid(waves, TRS_BLOCK / TRS_WAVE);
  hipLaunchKernelGGL((rows_scatter_add_kernel<LPR>), dim3(grid), dim3(TRS_BLOCK), 0, s, table, n_rows, D, idx, ib,
                     vals, ld, n, alpha, err);
}

static void dispatch_scatter(float* table, int64_t n_rows, int D, const void* idx, int ib, const float* vals,
                             int64_t ld, int64_t n, float alpha, int32_t* err, hipStream_t s) {
  if (D >= 64) launch_scatter<64>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
  else if (D > 16) launch_scatter<32>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
  else if (D > 8) launch_scatter<16>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
  else if (D > 4) launch_scatter<8>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
  else if (D > 1) launch_scatter<4>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
  else launch_scatter<1>(table, n_rows, D, idx, ib, vals, ld, n, alpha, err, s);
}

// ---- fused SGD update of all R fields of a scorer -------------------------------------------------------------
struct SgdArgs {
  trs_tables T;
  trs_batch Bt;
  const float* grad_rows;  // (R,B,D)
  const float* grad_lin;   // (R,B)
  float lr;
  int has_lin_meta;  // FM: metadata fields own 1-wide tables too
  int f_begin;       // first field to apply (0: all; 3: metadata fields only — user / item rows went the presorted way)
};

__device__ __forceinline__ void field_lookup(const SgdArgs& a, int f, float*& tab, float*& lin, const void*& idx,
                                             int64_t& n_rows, int& stride, int& col) {
  const trs_tables& T = a.T;
  stride = 1;
  col = 0;
  if (f == 0) { tab = T.user; lin = T.user_lin; idx = a.Bt.user; n_rows = T.n_users; }
  else if (f == 1) { tab = T.item; lin = T.item_lin; idx = a.Bt.pos; n_rows = T.n_items; }
  else if (f == 2) { tab = T.item; lin = T.item_lin; idx = a.Bt.neg; n_rows = T.n_items; }
  else {
    const int m = (f - 3) >> 1;
    tab = T.meta[m];
    lin = a.has_lin_meta ? T.meta_lin[m] : nullptr;
    idx = ((f - 3) & 1) ? a.Bt.neg_meta : a.Bt.pos_meta;
    n_rows = T.n_meta[m];
    stride = T.M;
    col = m;
  }
}

template <int LPR>
__global__ __launch_bounds__(TRS_BLOCK) void score_sgd_update_kernel(const SgdArgs a) {
  constexpr int EPW = TRS_WAVE / LPR;
  const int D = a.T.D;
  const int R = 3 + 2 * a.T.M;
  const int64_t B = a.Bt.B;
  const int ib = a.Bt.idx_bytes;
  const int lane = threadIdx.x & 63;
  const int lir = lane % LPR;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t n = (int64_t)R * B;
  const int64_t e0 = (int64_t)a.f_begin * B;
  const int64_t niter = (n - e0 + EPW - 1) / EPW;
  const float alpha = -a.lr;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t e = e0 + it * EPW + lane / LPR;
    if (e >= n) continue;
    const int f = (int)(e / B);
    const int64_t t = e - (int64_t)f * B;
    float *tab, *lin;
    const void* idx;
    int64_t n_rows;
    int stride, col;
    field_lookup(a, f, tab, lin, idx, n_rows, stride, col);
    const int64_t row = trs_ld_idx(idx, ib, t * stride + col);
    if ((uint64_t)row >= (uint64_t)n_rows) {
      if (lir == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
      continue;
    }
    float* dst = tab + row * (int64_t)D;
    const float* src = a.grad_rows + e * (int64_t)D;
    for (int d = lir; d < D; d += LPR) atomicAdd(dst + d, alpha * src[d]);
    if (lir == 0 && lin) atomicAdd(lin + row, alpha * a.grad_lin[e]);
  }
}

// ---- coalescing optimisers: owner election + apply -----------------------------------------------------------
// mode 0: SparseAdam (torch/optim/sparse_adam.py + _functional.sparse_adam); mode 1: Adagrad sparse branch
// (torch/optim/adagrad.py _single_tensor_adagrad).
struct ApplyArgs {
  float* table;
  float* acc;
  float* s1;  // exp_avg | state_sum
  float* s2;  // exp_avg_sq | unused
  int32_t* stamp;
  int64_t n_rows;
  int D;
  const void* idx;
  int ib;
  int64_t n;
  int32_t step_id;
  float lr_eff;  // SparseAdam: lr*sqrt(1-b2^t)/(1-b1^t); Adagrad: clr
  float beta1, beta2, eps;
};

template <int LPR, int MODE>
__global__ __launch_bounds__(TRS_BLOCK) void rows_apply_kernel(const ApplyArgs a) {
  constexpr int EPW = TRS_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int lir = lane % LPR;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t niter = (a.n + EPW - 1) / EPW;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t e = it * EPW + lane / LPR;
    int64_t row = -1;
    if (e < a.n) {
      row = trs_ld_idx(a.idx, a.ib, e);
      if ((uint64_t)row >= (uint64_t)a.n_rows) row = -1;
    }
    int owner = 0;
    if (row >= 0 && lir == 0) owner = atomicExch(a.stamp + row, a.step_id) != a.step_id;
    owner = __shfl(owner, (lane / LPR) * LPR, 64);
    if (!owner) continue;
    const int64_t base = row * (int64_t)a.D;
    for (int d = lir; d < a.D; d += LPR) {
      const float g = a.acc[base + d];
      a.acc[base + d] = 0.f;
      float w = a.table[base + d];
      if (MODE == 0) {
        const float m0 = a.s1[base + d], v0 = a.s2[base + d];
        const float dm = (g - m0) * (1.0f - a.beta1);
        const float dv = (g * g - v0) * (1.0f - a.beta2);
        a.s1[base + d] = m0 + dm;
        a.s2[base + d] = v0 + dv;
        const float numer = dm + m0;
        const float denom = sqrtf(dv + v0) + a.eps;
        w += -a.lr_eff * (numer / denom);
      } else {
        const float s = a.s1[base + d] + g * g;
        a.s1[base + d] = s;
        w += -a.lr_eff * (g / (sqrtf(s) + a.eps));
      }
      a.table[base + d] = w;
    }
  }
}

template <int MODE>
static void dispatch_apply(const ApplyArgs& a, hipStream_t s) {
#define TRS_APPLY(L)                                                                                   \
  {                                                                                                    \
    constexpr int EPW = TRS_WAVE / L;                                                                  \
    const int grid = trs_grid((a.n + EPW - 1) / EPW, TRS_BLOCK / TRS_WAVE);                            \
    hipLaunchKernelGGL((rows_apply_kernel<L, MODE>), dim3(grid), dim3(TRS_BLOCK), 0, s, a);            \
  }
  if (a.D >= 64) TRS_APPLY(64)
  else if (a.D > 16) TRS_APPLY(32)
  else if (a.D > 8) TRS_APPLY(16)
  else if (a.D > 4) TRS_APPLY(8)
  else if (a.D > 1) TRS_APPLY(4)
  else TRS_APPLY(1)
#undef TRS_APPLY
}

}  // namespace

extern "C" int trs_rows_scatter_add(float* table_dev, int64_t n_rows, int32_t D, const void* idx_dev,
                                    int32_t idx_bytes, const float* vals_dev, int64_t ld, int64_t n, float alpha,
                                    int32_t* err_flag_dev, void* stream) {
  TRS_REQUIRE(table_dev && n_rows > 0 && D > 0, "trs_rows_scatter_add: bad table (n_rows=%lld, D=%d)",
              (long long)n_rows, D);
  TRS_REQUIRE(idx_bytes == 4 || idx_bytes == 8, "trs_rows_scatter_add: idx_bytes must be 4 or 8");
  TRS_REQUIRE(n >= 0 && ld >= D, "trs_rows_scatter_add: need n >= 0 and ld >= D");
  if (n == 0) return TRS_OK;
  TRS_REQUIRE(idx_dev && vals_dev, "trs_rows_scatter_add: idx/vals is NULL");
  dispatch_scatter(table_dev, n_rows, D, idx_dev, idx_bytes, vals_dev, ld, n, alpha, err_flag_dev,
                   (hipStream_t)stream);
  TRS_CHECK_LAUNCH("rows_scatter_add_kernel");
  return TRS_OK;
}

int trs_launch_sgd_fields(int net, const trs_tables* tables, const trs_batch* batch, const float* grad_rows_dev,
                          const float* grad_lin_dev, float lr, int f_begin, void* stream);

extern "C" int trs_score_sgd_update(int net, const trs_tables* tables, const trs_batch* batch,
                                    const float* grad_rows_dev, const float* grad_lin_dev, float lr, void* stream) {
  return trs_launch_sgd_fields(net, tables, batch, grad_rows_dev, grad_lin_dev, lr, 0, stream);
}

// Atomic SGD scatter of the staged gradient fields [f_begin, R): f_begin = 3 applies the metadata fields only
// (csrc/fast_step.hip: metadata scorers on the presorted step).
int trs_launch_sgd_fields(int net, const trs_tables* tables, const trs_batch* batch, const float* grad_rows_dev,
                          const float* grad_lin_dev, float lr, int f_begin, void* stream) {
  TRS_REQUIRE(tables && batch, "trs_score_sgd_update: tables/batch is NULL");
  TRS_REQUIRE(net == TRS_NET_LINEAR || net == TRS_NET_FM, "trs_score_sgd_update: bad net");
  TRS_REQUIRE(tables->M >= 0 && tables->M <= TRS_MAX_META, "trs_score_sgd_update: bad M");
  TRS_REQUIRE(batch->idx_bytes == 4 || batch->idx_bytes == 8, "trs_score_sgd_update: idx_bytes must be 4 or 8");
  if (batch->B == 0) return TRS_OK;
  TRS_REQUIRE(batch->user && batch->pos && batch->neg, "trs_score_sgd_update: ids are NULL");
  TRS_REQUIRE(tables->M == 0 || (batch->pos_meta && batch->neg_meta), "trs_score_sgd_update: metadata ids are NULL");
  TRS_REQUIRE(grad_rows_dev && grad_lin_dev, "trs_score_sgd_update: staging arrays are NULL");
  SgdArgs a;
  a.T = *tables;
  a.Bt = *batch;
  a.grad_rows = grad_rows_dev;
  a.grad_lin = grad_lin_dev;
  a.lr = lr;
  a.has_lin_meta = net == TRS_NET_FM;
  a.f_begin = f_begin;
  const int D = tables->D;
  const int64_t n = (int64_t)(3 + 2 * tables->M - f_begin) * batch->B;
  if (n <= 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
#define TRS_SGD(L)                                                                                  \
  {                                                                                                 \
    constexpr int EPW = TRS_WAVE / L;                                                               \
    const int grid = trs_grid((n + EPW - 1) / EPW, TRS_BLOCK / TRS_WAVE);                           \
    hipLaunchKernelGGL((score_sgd_update_kernel<L>), dim3(grid), dim3(TRS_BLOCK), 0, s, a);         \
  }
  if (D >= 64) TRS_SGD(64)
  else if (D > 16) TRS_SGD(32)
  else if (D > 8) TRS_SGD(16)
  else if (D > 4) TRS_SGD(8)
  else if (D > 1) TRS_SGD(4)
  else TRS_SGD(1)
#undef TRS_SGD
  TRS_CHECK_LAUNCH("score_sgd_update_kernel");
  return TRS_OK;
}

static int check_apply(const char* who, float* table, float* acc, float* s1, int32_t* stamp, int64_t n_rows,
                       int32_t D, const void* idx, int32_t ib, int64_t n, int32_t step_id) {
  TRS_REQUIRE(table && acc && s1 && stamp, "%s: table/acc/state/stamp is NULL", who);
  TRS_REQUIRE(n_rows > 0 && D > 0, "%s: bad table shape", who);
  TRS_REQUIRE(ib == 4 || ib == 8, "%s: idx_bytes must be 4 or 8", who);
  TRS_REQUIRE(n >= 0, "%s: negative n", who);
  TRS_REQUIRE(n == 0 || idx, "%s: idx is NULL", who);
  TRS_REQUIRE(step_id != 0, "%s: step_id must not be 0", who);
  return TRS_OK;
}

extern "C" int trs_rows_apply_sparse_adam(float* table_dev, float* acc_dev, float* exp_avg_dev,
                                          float* exp_avg_sq_dev, int32_t* stamp_dev, int64_t n_rows, int32_t D,
                                          const void* idx_dev, int32_t idx_bytes, int64_t n, int32_t step_id,
                                          float lr, float beta1, float beta2, float eps, int64_t step_count,
                                          void* stream) {
  int rc = check_apply("trs_rows_apply_sparse_adam", table_dev, acc_dev, exp_avg_dev, stamp_dev, n_rows, D, idx_dev,
                       idx_bytes, n, step_id);
  if (rc) return rc;
  TRS_REQUIRE(exp_avg_sq_dev, "trs_rows_apply_sparse_adam: exp_avg_sq is NULL");
  TRS_REQUIRE(step_count >= 1, "trs_rows_apply_sparse_adam: step_count must be >= 1");
  if (n == 0) return TRS_OK;
  // step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in double like the Python floats of torch
  const double bc1 = 1.0 - pow((double)beta1, (double)step_count);
  const double bc2 = 1.0 - pow((double)beta2, (double)step_count);
  ApplyArgs a = {table_dev, acc_dev, exp_avg_dev, exp_avg_sq_dev, stamp_dev, n_rows, D, idx_dev, idx_bytes, n,
                 step_id, (float)((double)lr * sqrt(bc2) / bc1), beta1, beta2, eps};
  dispatch_apply<0>(a, (hipStream_t)stream);
  TRS_CHECK_LAUNCH("rows_apply_kernel<adam>");
  return TRS_OK;
}

extern "C" int trs_rows_apply_adagrad(float* table_dev, float* acc_dev, float* state_sum_dev, int32_t* stamp_dev,
                                      int64_t n_rows, int32_t D, const void* idx_dev, int32_t idx_bytes, int64_t n,
                                      int32_t step_id, float clr, float eps, void* stream) {
  int rc = check_apply("trs_rows_apply_adagrad", table_dev, acc_dev, state_sum_dev, stamp_dev, n_rows, D, idx_dev,
                       idx_bytes, n, step_id);
  if (rc) return rc;
  if (n == 0) return TRS_OK;
  ApplyArgs a = {table_dev, acc_dev, state_sum_dev, nullptr, stamp_dev, n_rows, D, idx_dev, idx_bytes, n,
                 step_id, clr, 0.f, 0.f, eps};
  dispatch_apply<1>(a, (hipStream_t)stream);
  TRS_CHECK_LAUNCH("rows_apply_kernel<adagrad>");
  return TRS_OK;
}
