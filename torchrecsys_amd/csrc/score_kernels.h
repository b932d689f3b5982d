// score_kernels.h — fused positive+negative scoring kernels for the Linear and FM scorers.
//
// Math: SURVEY.md App. A.2 / A.3 (Linear: collaborative/linear.py:54-80, FM: collaborative/fm.py:60-101 of the
// reference); hinge: helper/loss.py:5-9.
//
// Mapping (CDNA4, wave = 64): one aligned group of G lanes owns one triple (user, pos, neg); a lane holds
// K chunks of VEC consecutive floats of every row it touches (chunk k covers elements (k*G + lig)*VEC ..+VEC-1), so a
// row of D <= G*K*VEC floats is fetched by ONE 16-byte-per-lane wave instruction per chunk and a wave streams 64/G
// triples at a time.  The user row is loaded once and used by both passes.  Reductions over D are xor-shuffles inside
// the group.  All rows are read from the pre-update tables; nothing here writes a table.
#pragma once
#include "trs_common.h"
#include "opt_rows.h"

namespace trs {

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<2> {
  using type = float2;
};
template <>
struct VecT<1> {
  using type = float;
};

// a K x VEC register tile of one row
template <int VEC, int K>
struct RowReg {
  float v[K * VEC];
};

// Branch-free: hipcc closes every conditional block that contains a load with `s_waitcnt vmcnt(0)`, which serialises
// the three row gathers of a triple (and drains anything prefetched).  Lanes past the row's end load the row's first
// chunk instead and are zeroed by a multiply with a 0/1 mask (a select would let the optimiser sink the load back under
// the condition); FULL = the lane group covers the row exactly (D == G*K*VEC): no mask at all.
// NT: nontemporal load (rows that are read once and should not displace what is re-read: the read-only scoring pass).
template <int VEC, int G, int K, bool FULL = false, bool NT = false>
__device__ __forceinline__ void row_load(RowReg<VEC, K>& r, const float* __restrict__ tab, int64_t row, int D,
                                         int lig) {
  const float* p = tab + row * (int64_t)D;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = (k * G + lig) * VEC;
    if (FULL) {
      using VT = float __attribute__((ext_vector_type(VEC)));
      typename VecT<VEC>::type x;
      if (NT) {
        const VT t = __builtin_nontemporal_load(reinterpret_cast<const VT*>(p + e));
        x = __builtin_bit_cast(typename VecT<VEC>::type, t);
      } else {
        x = *reinterpret_cast<const typename VecT<VEC>::type*>(p + e);
      }
      const float* xs = reinterpret_cast<const float*>(&x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) r.v[k * VEC + c] = xs[c];
    } else {
      const bool act = e < D;
      const float m = act ? 1.0f : 0.0f;
      typename VecT<VEC>::type x = *reinterpret_cast<const typename VecT<VEC>::type*>(p + (act ? e : 0));
      const float* xs = reinterpret_cast<const float*>(&x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) r.v[k * VEC + c] = xs[c] * m;
    }
  }
}

template <int VEC, int G, int K, bool NT = false>
__device__ __forceinline__ void row_store(const RowReg<VEC, K>& r, float* __restrict__ dst, int D, int lig) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = (k * G + lig) * VEC;
    if (e < D) {
      typename VecT<VEC>::type x;
      float* xs = reinterpret_cast<float*>(&x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) xs[c] = r.v[k * VEC + c];
      if (NT) {
        using VT = float __attribute__((ext_vector_type(VEC)));
        __builtin_nontemporal_store(__builtin_bit_cast(VT, x), reinterpret_cast<VT*>(dst + e));
      } else {
        *reinterpret_cast<typename VecT<VEC>::type*>(dst + e) = x;
      }
    }
  }
}

struct ScoreArgs {
  trs_tables T;
  trs_batch Bt;
  float inv_B;
  int loss;  // TRS_LOSS_HINGE | TRS_LOSS_BPR (MODE 1 without upstream gradients, MODE 2, meta_stage_kernel)
  float* pos_score;
  float* neg_score;
  float* loss_sum;
  int32_t* auc_count;
  float* grad_rows;   // (R,B,D)
  float* grad_lin;    // (R,B)
  const float* gpos;  // upstream d loss / d score (B,) or NULL -> hinge
  const float* gneg;
  // predict mode (trs_score_all_items): every triple is (iota_user, iota_item0 + t); metadata ids of item p are
  // iota_item_meta[p*M + m] (int32).  iota_user < 0 = off.
  int64_t iota_user;
  int64_t iota_item0;
  const int32_t* iota_item_meta;
  // MODE 2 (metadata scorers on the presorted step, csrc/fast_step.hip): metadata ids are looked up from the item ->
  // metadata table; the kernel writes gz (2,B), the rows the sorted item update needs (xstage: FM (2,B,D) = the per-pass
  // field sums S+ / S-, Linear (B,D) = the user row), applies the SGD update of users referenced once in the batch in
  // place (udup_pos[t] == 0) or stages their gradient (du), stages the metadata fields' gradients in grad_rows /
  // grad_lin (fields 3..) and their ids in meta_ids_out (2,B,M) for the atomic scatter of those small tables.
  const int32_t* item_meta_tab;  // (n_items, M) int32
  float* gz;
  float* du;
  float* xstage;
  const uint8_t* udup_pos;
  float lr;
  int32_t* meta_ids_out;
  OptArgs o;  // meta_stage_kernel (fast_step.hip): update rule of the in-place user update (kind OPT_SGD: lr above)
};

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// One pass (positive or negative) of a scorer for one triple, given the user row.
//   Ssum  : FM: sum over fields of the rows (incl. user);  Linear: item + sum of metadata rows
//   score : the pass's score (FM: sigmoid(z))
template <int NET, int VEC, int G, int K>
__device__ __forceinline__ float pass_forward(const trs_tables& T, const RowReg<VEC, K>& u, float u_lin,
                                              int64_t item, const void* meta, int idx_bytes, int64_t t, bool valid,
                                              int lig, RowReg<VEC, K>& it, RowReg<VEC, K>& Ssum, float& it_lin,
                                              float& lin_sum, bool& ok) {
  constexpr int N = K * VEC;
  const int D = T.D;
  row_load<VEC, G, K>(it, T.item, item, D, lig);
  it_lin = T.item_lin ? T.item_lin[item] : 0.f;
  float sq[N];  // FM: sum over fields of v^2
  if (NET == TRS_NET_FM) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      Ssum.v[n] = u.v[n] + it.v[n];
      sq[n] = u.v[n] * u.v[n] + it.v[n] * it.v[n];
    }
    lin_sum = u_lin + it_lin;
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) Ssum.v[n] = it.v[n];
    lin_sum = 0.f;
  }
  for (int m = 0; m < T.M; ++m) {
    int64_t mid = trs_ld_idx(meta, idx_bytes, t * T.M + m);  // t is already clamped into the batch by the caller
    if ((uint64_t)mid >= (uint64_t)T.n_meta[m]) {
      ok = false;
      mid = 0;
    }
    RowReg<VEC, K> mr;
    row_load<VEC, G, K>(mr, T.meta[m], mid, D, lig);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      Ssum.v[n] += mr.v[n];
      if (NET == TRS_NET_FM) sq[n] += mr.v[n] * mr.v[n];
    }
    if (NET == TRS_NET_FM) lin_sum += T.meta_lin[m][mid];
  }
  float part = 0.f;
  if (NET == TRS_NET_FM) {
#pragma unroll
    for (int n = 0; n < N; ++n) part += Ssum.v[n] * Ssum.v[n] - sq[n];
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) part += u.v[n] * Ssum.v[n];
  }
  const float red = trs_group_sum<G>(part);
  if (NET == TRS_NET_FM) return sigmoidf_(lin_sum + 0.5f * red);
  return (red + u_lin) + it_lin;  // (dot + user_bias) + item_bias, linear.py:78
}

// MODE 0: scores only.  MODE 1: scores + hinge/upstream grads -> staged per-triple gradient rows.
template <int NET, int VEC, int G, int K, int MODE>
__global__ __launch_bounds__(TRS_BLOCK) void score_kernel(const ScoreArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;  // triples per wave per iteration
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.Bt.B;
  const int ib = a.Bt.idx_bytes;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const bool has_neg = a.Bt.neg != nullptr;

  float loss_acc = 0.f;
  int auc_acc = 0;

  const int64_t niter = (B + TPW - 1) / TPW;
  for (int64_t it_ = wave; it_ < niter; it_ += nwave) {
    const int64_t t = it_ * TPW + lane / G;
    const bool valid = t < B;
    bool ok = true;
    const bool iota = a.iota_user >= 0;
    const int64_t tc = valid ? t : 0;  // loads stay unconditional (a conditional load costs an s_waitcnt vmcnt(0))
    int64_t uid, pid, nid = 0;
    if (iota) {
      uid = a.iota_user;
      pid = a.iota_item0 + tc;
    } else {
      uid = trs_ld_idx(a.Bt.user, ib, tc);
      pid = trs_ld_idx(a.Bt.pos, ib, tc);
      if (has_neg) nid = trs_ld_idx(a.Bt.neg, ib, tc);
    }
    if ((uint64_t)uid >= (uint64_t)T.n_users) { ok = false; uid = 0; }
    if ((uint64_t)pid >= (uint64_t)T.n_items) { ok = false; pid = 0; }
    if ((uint64_t)nid >= (uint64_t)T.n_items) { ok = false; nid = 0; }

    RowReg<VEC, K> u, pi, ni, Sp, Sn;
    row_load<VEC, G, K>(u, T.user, uid, D, lig);
    const float u_lin = T.user_lin ? T.user_lin[uid] : 0.f;
    float pi_lin, ni_lin = 0.f, lin_p, lin_n = 0.f;
    // MODE 2: metadata ids from the item -> metadata table (indexed by the item id) unless the presort wrote them out
    const bool TAB = MODE == 2 && a.Bt.pos_meta == nullptr;
    const float sp = pass_forward<NET, VEC, G, K>(
        T, u, u_lin, pid, iota ? (const void*)a.iota_item_meta : (TAB ? (const void*)a.item_meta_tab : a.Bt.pos_meta),
        (iota || TAB) ? 4 : ib, (iota || TAB) ? pid : tc, valid, lig, pi, Sp, pi_lin, lin_p, ok);
    float sn = 0.f;
    if (has_neg)
      sn = pass_forward<NET, VEC, G, K>(T, u, u_lin, nid, TAB ? (const void*)a.item_meta_tab : a.Bt.neg_meta,
                                        TAB ? 4 : ib, TAB ? nid : tc, valid, lig, ni, Sn, ni_lin, lin_n, ok);
    if (valid && !ok && lig == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
    const bool live = valid && ok;

    if (valid && lig == 0) {
      if (a.pos_score) a.pos_score[t] = live ? sp : 0.f;
      if (a.neg_score && has_neg) a.neg_score[t] = live ? sn : 0.f;
    }

    if (MODE == 1) {
      // upstream gradients of the two scores
      float gp, gn;
      if (a.gpos) {
        gp = live ? a.gpos[t] : 0.f;
        gn = live ? a.gneg[t] : 0.f;
      } else {
        float lval, dneg;  // helper/loss.py:7 (hinge: clamp(min=0) passes the gradient at h == 0) or BPR
        trs_pair_loss(a.loss, sp, sn, lval, dneg);
        const float act = live ? dneg : 0.f;
        gp = -act * a.inv_B;
        gn = act * a.inv_B;
        if (live && lig == 0) {
          loss_acc += lval;
          auc_acc += (sp > sn) ? 1 : 0;
        }
      }
      if (NET == TRS_NET_FM) {  // through the sigmoid: g * s * (1 - s)
        gp = gp * ((1.0f - sp) * sp);
        gn = gn * ((1.0f - sn) * sn);
      }
      if (valid) {
        const int M = T.M;
        float* gr = a.grad_rows;
        const int64_t BD = B * (int64_t)D;
        RowReg<VEC, K> g;
        // field 0: user.  FM: gp*(Sp-u) + gn*(Sn-u) (= gp*i + gn*j when M == 0); Linear: gp*Ip + gn*In
        if (NET == TRS_NET_FM) {
          if (M == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) g.v[n] = gp * pi.v[n] + gn * ni.v[n];
          } else {
#pragma unroll
            for (int n = 0; n < N; ++n) g.v[n] = gp * (Sp.v[n] - u.v[n]) + gn * (Sn.v[n] - u.v[n]);
          }
        } else {
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gp * Sp.v[n] + gn * Sn.v[n];
        }
        row_store<VEC, G, K>(g, gr + 0 * BD + t * (int64_t)D, D, lig);
        // field 1 / 2: pos / neg item.  FM: g*(S - item) (= g*u when M == 0); Linear: g*u
        if (NET == TRS_NET_FM && M != 0) {
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gp * (Sp.v[n] - pi.v[n]);
        } else {
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gp * u.v[n];
        }
        row_store<VEC, G, K>(g, gr + 1 * BD + t * (int64_t)D, D, lig);
        if (NET == TRS_NET_FM && M != 0) {
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gn * (Sn.v[n] - ni.v[n]);
        } else {
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gn * u.v[n];
        }
        row_store<VEC, G, K>(g, gr + 2 * BD + t * (int64_t)D, D, lig);
        // metadata fields (rows re-read: they are L1/L2-hot from the forward part)
        for (int m = 0; m < M; ++m) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const float gs = s ? gn : gp;
            const RowReg<VEC, K>& S = s ? Sn : Sp;
            if (NET == TRS_NET_FM) {
              int64_t mid = trs_ld_idx(s ? a.Bt.neg_meta : a.Bt.pos_meta, ib, t * M + m);
              if ((uint64_t)mid >= (uint64_t)T.n_meta[m]) mid = 0;
              RowReg<VEC, K> mr;
              row_load<VEC, G, K>(mr, T.meta[m], mid, D, lig);
#pragma unroll
              for (int n = 0; n < N; ++n) g.v[n] = gs * (S.v[n] - mr.v[n]);
            } else {
#pragma unroll
              for (int n = 0; n < N; ++n) g.v[n] = gs * u.v[n];
            }
            row_store<VEC, G, K>(g, gr + (int64_t)(3 + 2 * m + s) * BD + t * (int64_t)D, D, lig);
          }
        }
        // 1-wide terms: FM linear_* / Linear biases
        if (lig == 0 && a.grad_lin) {
          float* gl = a.grad_lin;
          gl[0 * B + t] = gp + gn;
          gl[1 * B + t] = gp;
          gl[2 * B + t] = gn;
          for (int m = 0; m < M; ++m) {  // Linear has no 1-wide metadata tables: those fields stay 0
            gl[(int64_t)(3 + 2 * m) * B + t] = NET == TRS_NET_FM ? gp : 0.f;
            gl[(int64_t)(4 + 2 * m) * B + t] = NET == TRS_NET_FM ? gn : 0.f;
          }
        }
      }
    }
    if (MODE == 2) {
      float lval, dneg;
      trs_pair_loss(a.loss, sp, sn, lval, dneg);
      const float act = live ? dneg : 0.f;
      float gp = -act * a.inv_B, gn = act * a.inv_B;
      if (live && lig == 0) loss_acc += lval;
      if (NET == TRS_NET_FM) {
        gp = gp * ((1.0f - sp) * sp);
        gn = gn * ((1.0f - sn) * sn);
      }
      if (valid) {
        const int M = T.M;
        const int64_t BD = B * (int64_t)D;
        RowReg<VEC, K> g;
        // rows the sorted item update multiplies by its coefficients: FM needs S of the reference's own pass (the
        // gradient of a field row v is g*(S - v), and v is the row the run kernel holds), Linear the user row
        if (NET == TRS_NET_FM) {
          row_store<VEC, G, K>(Sp, a.xstage + t * (int64_t)D, D, lig);
          row_store<VEC, G, K>(Sn, a.xstage + BD + t * (int64_t)D, D, lig);
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gp * (Sp.v[n] - u.v[n]) + gn * (Sn.v[n] - u.v[n]);
        } else {
          row_store<VEC, G, K>(u, a.xstage + t * (int64_t)D, D, lig);
#pragma unroll
          for (int n = 0; n < N; ++n) g.v[n] = gp * Sp.v[n] + gn * Sn.v[n];
        }
        if (a.udup_pos[t] || !live) {
          row_store<VEC, G, K>(g, a.du + t * (int64_t)D, D, lig);
        } else {
          RowReg<VEC, K> un;
#pragma unroll
          for (int n = 0; n < N; ++n) un.v[n] = u.v[n] + (-a.lr) * g.v[n];
          row_store<VEC, G, K>(un, T.user + uid * (int64_t)D, D, lig);
          if (lig == 0 && T.user_lin) T.user_lin[uid] = u_lin + (-a.lr) * (gp + gn);
        }
        // metadata fields: staged gradients (as MODE 1 stages them) + their ids — unless the columns have sorted runs
        // of their own (grad_rows == NULL: they read xstage like the item runs)
        for (int m = 0; m < (a.grad_rows ? M : 0); ++m) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const float gs = s ? gn : gp;
            const RowReg<VEC, K>& S = s ? Sn : Sp;
            int64_t mid = a.Bt.pos_meta ? trs_ld_idx(s ? a.Bt.neg_meta : a.Bt.pos_meta, ib, t * M + m)
                                        : (int64_t)a.item_meta_tab[(s ? nid : pid) * M + m];
            if ((uint64_t)mid >= (uint64_t)T.n_meta[m]) mid = 0;
            if (NET == TRS_NET_FM) {
              RowReg<VEC, K> mr;
              row_load<VEC, G, K>(mr, T.meta[m], mid, D, lig);
#pragma unroll
              for (int n = 0; n < N; ++n) g.v[n] = gs * (S.v[n] - mr.v[n]);
            } else {
#pragma unroll
              for (int n = 0; n < N; ++n) g.v[n] = gs * u.v[n];
            }
            row_store<VEC, G, K>(g, a.grad_rows + (int64_t)(3 + 2 * m + s) * BD + t * (int64_t)D, D, lig);
            if (lig == 0) {
              a.meta_ids_out[((int64_t)s * B + t) * M + m] = (int32_t)mid;  // a dead triple carries zero gradients
              if (a.grad_lin) a.grad_lin[(int64_t)(3 + 2 * m + s) * B + t] = NET == TRS_NET_FM ? gs : 0.f;
            }
          }
        }
        if (lig == 0) {
          a.gz[t] = gp;
          a.gz[B + t] = gn;
        }
      }
    }

  }

  if (MODE != 0 && a.loss_sum) {
    __shared__ float s_loss[TRS_BLOCK / TRS_WAVE];
    __shared__ int s_auc[TRS_BLOCK / TRS_WAVE];
    const float wl = trs_wave_sum(loss_acc);
    const int wa = trs_wave_sum_i(auc_acc);
    if (lane == 0) {
      s_loss[threadIdx.x >> 6] = wl;
      s_auc[threadIdx.x >> 6] = wa;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float L = 0.f;
      int A = 0;
#pragma unroll
      for (int w = 0; w < TRS_BLOCK / TRS_WAVE; ++w) {
        L += s_loss[w];
        A += s_auc[w];
      }
      if (L != 0.f) atomicAdd(a.loss_sum, L);
      if (A != 0 && a.auc_count) atomicAdd(a.auc_count, A);
    }
  }
}

// ---------------------------------------------------------------------------------------- dispatch
// Chooses (VEC, G, K) for a row width D; returns false when D is unsupported (D > 1024 or D < 1).
struct RowCfg {
  int vec, g, k;
};
static inline bool pick_row_cfg(int D, RowCfg& c) {
  if (D < 1 || D > 1024) return false;
  if (D % 4 == 0) {
    const int chunks = D / 4;
    c.vec = 4;
    if (chunks <= 64) {
      int g = 2;  // smallest instantiated group
      while (g < chunks) g <<= 1;
      c.g = g;
      c.k = 1;
    } else {
      c.g = 64;
      c.k = chunks <= 128 ? 2 : 4;
    }
    return true;
  }
  c.vec = 1;
  if (D <= 4) { c.g = 4; c.k = 1; }
  else if (D <= 16) { c.g = 16; c.k = 1; }
  else if (D <= 64) { c.g = 64; c.k = 1; }
  else if (D <= 256) { c.g = 64; c.k = 4; }
  else return false;  // odd D > 256 is not instantiated
  return true;
}

template <int NET, int MODE>
static inline int launch_score(const ScoreArgs& a, hipStream_t s) {
  RowCfg c;
  if (!pick_row_cfg(a.T.D, c)) {
    trs_set_error("unsupported n_factors D=%d (need 1..1024; D %% 4 != 0 only up to 256)", a.T.D);
    return TRS_E_ARG;
  }
  const int64_t B = a.Bt.B;
  if (B == 0) return TRS_OK;
  const int tpw = TRS_WAVE / c.g;
  const int64_t waves = (B + tpw - 1) / tpw;
  const int grid = trs_grid(waves, TRS_BLOCK / TRS_WAVE);
#define TRS_CASE(V, GG, KK)                                                                        \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                      \
    hipLaunchKernelGGL((score_kernel<NET, V, GG, KK, MODE>), dim3(grid), dim3(TRS_BLOCK), 0, s, a); \
    TRS_CHECK_LAUNCH("score_kernel");                                                              \
    return TRS_OK;                                                                                 \
  }
  TRS_CASE(4, 2, 1)
  TRS_CASE(4, 4, 1)
  TRS_CASE(4, 8, 1)
  TRS_CASE(4, 16, 1)
  TRS_CASE(4, 32, 1)
  TRS_CASE(4, 64, 1)
  TRS_CASE(4, 64, 2)
  TRS_CASE(4, 64, 4)
  TRS_CASE(1, 4, 1)
  TRS_CASE(1, 16, 1)
  TRS_CASE(1, 64, 1)
  TRS_CASE(1, 64, 4)
#undef TRS_CASE
  trs_set_error("internal: no kernel for D=%d", a.T.D);
  return TRS_E_ARG;
}

}  // namespace trs
