// opt_rows.h — per-row optimiser rules shared by the fused step kernels (fast_step.hip, presort.hip).
//   SGD       W += -lr * G                                        (torch/optim/sgd.py, momentum = 0)
//   ADAM      lazy / SparseAdam on the rows present in the batch   (torch/optim/_functional.py sparse_adam):
//             m += (1-b1)(G-m); v += (1-b2)(G^2-v); W += -step_size * m / (sqrt(v) + eps),
//             step_size = lr * sqrt(1-b2^t) / (1-b1^t) evaluated on the host in double
//   ADAGRAD   sparse branch of torch/optim/adagrad.py:  sum += G^2;  W += -clr * G / (sqrt(sum) + eps)
// G is the COALESCED gradient of the row (all references of the step summed) — the adaptive rules are not linear in G,
// which is why they need the presorted runs.  Same arithmetic, operation by operation, as csrc/rows.hip.
#pragma once
#include "trs_common.h"

namespace trs {

enum { OPT_SGD = 0, OPT_ADAM = 1, OPT_ADAGRAD = 2 };

struct OptArgs {
  int kind;
  float lr_eff;  // SGD: lr; ADAM: step_size of this step; ADAGRAD: clr of this step
  float beta1, beta2, eps;
  // optimiser state, same shapes as the weight tables: s1 = exp_avg | sum, s2 = exp_avg_sq | unused
  float *user_s1, *user_s2, *item_s1, *item_s2;
  float *user_lin_s1, *user_lin_s2, *item_lin_s1, *item_lin_s2;
  // item runs cut at a chunk boundary (adaptive rules only): the pieces add their partial sums into gacc (all-zero
  // between steps), the head piece appends the row to the list, a small launch applies the rule once per listed row
  float* gacc;      // (n_items, D)
  float* gacc_lin;  // (n_items)
  int32_t* cut_rows;   // (capacity) row ids
  int32_t* cut_count;  // [2] counters, alternating by step parity
  int32_t cut_capacity;
};

template <int OPT>
__device__ __forceinline__ float opt_apply(float w, float g, float& s1, float& s2, const OptArgs& o) {
  if (OPT == OPT_ADAM) {
    const float m0 = s1, v0 = s2;
    const float dm = (g - m0) * (1.0f - o.beta1);
    const float dv = (g * g - v0) * (1.0f - o.beta2);
    s1 = m0 + dm;
    s2 = v0 + dv;
    const float numer = dm + m0;
    const float denom = sqrtf(dv + v0) + o.eps;
    return w + (-o.lr_eff) * (numer / denom);
  } else if (OPT == OPT_ADAGRAD) {
    const float s = s1 + g * g;
    s1 = s;
    return w + (-o.lr_eff) * (g / (sqrtf(s) + o.eps));
  } else {
    return w + (-o.lr_eff) * g;
  }
}

}  // namespace trs
