// score.hip — C-ABI entry points of the Linear / FM scorers (forward, forward+hinge+backward, backward).
#include "score_kernels.h"

using namespace trs;

static int check_tables(int net, const trs_tables* T, const char* who) {
  TRS_REQUIRE(T != nullptr, "%s: tables is NULL", who);
  TRS_REQUIRE(net == TRS_NET_LINEAR || net == TRS_NET_FM, "%s: net must be TRS_NET_LINEAR or TRS_NET_FM", who);
  TRS_REQUIRE(T->M >= 0 && T->M <= TRS_MAX_META, "%s: M=%d outside 0..%d", who, T->M, TRS_MAX_META);
  TRS_REQUIRE(T->user && T->item, "%s: user/item table is NULL", who);
  TRS_REQUIRE(T->user_lin && T->item_lin, "%s: 1-wide user/item table (bias / linear term) is NULL", who);
  TRS_REQUIRE(T->n_users > 0 && T->n_items > 0, "%s: empty user/item table", who);
  for (int m = 0; m < T->M; ++m) {
    TRS_REQUIRE(T->meta[m] && T->n_meta[m] > 0, "%s: metadata table %d is NULL/empty", who, m);
    if (net == TRS_NET_FM) TRS_REQUIRE(T->meta_lin[m], "%s: linear_metadata table %d is NULL", who, m);
  }
  return TRS_OK;
}

static int check_batch(const trs_tables* T, const trs_batch* b, bool need_neg, const char* who) {
  TRS_REQUIRE(b != nullptr, "%s: batch is NULL", who);
  TRS_REQUIRE(b->B >= 0, "%s: negative batch size", who);
  TRS_REQUIRE(b->idx_bytes == 4 || b->idx_bytes == 8, "%s: idx_bytes must be 4 or 8", who);
  if (b->B == 0) return TRS_OK;
  TRS_REQUIRE(b->user && b->pos, "%s: user/pos ids are NULL", who);
  if (need_neg) TRS_REQUIRE(b->neg, "%s: neg ids are NULL", who);
  if (T->M > 0) {
    TRS_REQUIRE(b->pos_meta, "%s: pos_meta ids are NULL but M=%d", who, T->M);
    if (b->neg) TRS_REQUIRE(b->neg_meta, "%s: neg_meta ids are NULL but M=%d", who, T->M);
  }
  return TRS_OK;
}

template <int MODE>
static int launch_net(int net, const ScoreArgs& a, hipStream_t s) {
  if (net == TRS_NET_FM) return launch_score<TRS_NET_FM, MODE>(a, s);
  return launch_score<TRS_NET_LINEAR, MODE>(a, s);
}

int trs_launch_pair_scores(int net, const trs::ScoreArgs* a, hipStream_t s);  // fast_step.hip

extern "C" int trs_score_forward(int net, const trs_tables* tables, const trs_batch* batch, float* pos_score_dev,
                                 float* neg_score_dev, void* stream) {
  int rc = check_tables(net, tables, "trs_score_forward");
  if (rc) return rc;
  rc = check_batch(tables, batch, false, "trs_score_forward");
  if (rc) return rc;
  TRS_REQUIRE(pos_score_dev, "trs_score_forward: pos_score is NULL");
  TRS_REQUIRE((batch->neg == nullptr) == (neg_score_dev == nullptr) || batch->B == 0,
              "trs_score_forward: neg ids and neg_score must both be given or both be NULL");
  ScoreArgs a = {};
  a.T = *tables;
  a.Bt = *batch;
  a.pos_score = pos_score_dev;
  a.neg_score = neg_score_dev;
  a.iota_user = -1;
  if (batch->B > 0) {  // no metadata, int32 ids, both passes: the software-pipelined form (fast_step.hip)
    rc = trs_launch_pair_scores(net, &a, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  return launch_net<0>(net, a, (hipStream_t)stream);
}

extern "C" int trs_score_fwd_bwd(int net, const trs_tables* tables, const trs_batch* batch, float inv_B,
                                 float* pos_score_dev, float* neg_score_dev, float* loss_sum_dev,
                                 int32_t* auc_count_dev, float* grad_rows_dev, float* grad_lin_dev, int32_t loss,
                                 void* stream) {
  TRS_REQUIRE(loss == TRS_LOSS_HINGE || loss == TRS_LOSS_BPR, "trs_score_fwd_bwd: bad loss kind");
  int rc = check_tables(net, tables, "trs_score_fwd_bwd");
  if (rc) return rc;
  rc = check_batch(tables, batch, true, "trs_score_fwd_bwd");
  if (rc) return rc;
  TRS_REQUIRE(grad_rows_dev && grad_lin_dev, "trs_score_fwd_bwd: staging arrays are NULL");
  TRS_REQUIRE(loss_sum_dev, "trs_score_fwd_bwd: loss_sum is NULL");
  ScoreArgs a = {};
  a.T = *tables;
  a.Bt = *batch;
  a.inv_B = inv_B;
  a.loss = loss;
  a.pos_score = pos_score_dev;
  a.neg_score = neg_score_dev;
  a.loss_sum = loss_sum_dev;
  a.auc_count = auc_count_dev;
  a.grad_rows = grad_rows_dev;
  a.grad_lin = grad_lin_dev;
  a.iota_user = -1;
  return launch_net<1>(net, a, (hipStream_t)stream);
}

extern "C" int trs_score_backward(int net, const trs_tables* tables, const trs_batch* batch, const float* gpos_dev,
                                  const float* gneg_dev, float* grad_rows_dev, float* grad_lin_dev, void* stream) {
  int rc = check_tables(net, tables, "trs_score_backward");
  if (rc) return rc;
  rc = check_batch(tables, batch, true, "trs_score_backward");
  if (rc) return rc;
  TRS_REQUIRE(gpos_dev && gneg_dev, "trs_score_backward: upstream gradients are NULL");
  TRS_REQUIRE(grad_rows_dev && grad_lin_dev, "trs_score_backward: staging arrays are NULL");
  ScoreArgs a = {};
  a.T = *tables;
  a.Bt = *batch;
  a.gpos = gpos_dev;
  a.gneg = gneg_dev;
  a.grad_rows = grad_rows_dev;
  a.grad_lin = grad_lin_dev;
  a.iota_user = -1;
  return launch_net<1>(net, a, (hipStream_t)stream);
}

extern "C" int trs_score_all_items(int net, const trs_tables* tables, int64_t user_id, int64_t item0, int64_t n,
                                   const int32_t* item_meta_dev, float* score_out_dev, void* stream) {
  int rc = check_tables(net, tables, "trs_score_all_items");
  if (rc) return rc;
  TRS_REQUIRE(user_id >= 0 && user_id < tables->n_users, "trs_score_all_items: user_id %lld out of range [0,%lld)",
              (long long)user_id, (long long)tables->n_users);
  TRS_REQUIRE(item0 >= 0 && n >= 0 && item0 + n <= tables->n_items,
              "trs_score_all_items: items [%lld,%lld) outside [0,%lld)", (long long)item0, (long long)(item0 + n),
              (long long)tables->n_items);
  TRS_REQUIRE(tables->M == 0 || item_meta_dev, "trs_score_all_items: item_meta is NULL but M=%d", tables->M);
  TRS_REQUIRE(score_out_dev, "trs_score_all_items: score_out is NULL");
  ScoreArgs a = {};
  a.T = *tables;
  a.Bt.B = n;
  a.Bt.idx_bytes = 4;
  a.pos_score = score_out_dev;
  a.iota_user = user_id;
  a.iota_item0 = item0;
  a.iota_item_meta = item_meta_dev;
  return launch_net<0>(net, a, (hipStream_t)stream);
}
