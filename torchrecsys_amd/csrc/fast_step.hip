// fast_step.hip — the specialised SGD training step of the Linear / FM scorers without metadata (M = 0), and the
// C-side step loop (one C call runs many steps: no per-step host work beyond four kernel launches).
//
// Exact reference semantics (every gradient of a step is taken at the PRE-update tables, reference model.py:188-200)
// with the least staging that allows it:
//   K1  fwd_stage    reads u, i, j rows (+1-wide terms) once — software-pipelined: the rows of the group's next triple
//                    and the ids of the one after are in flight while the current triple is reduced; [optionally]
//                    derives the batch from the resident interaction stream (Feistel shuffle + Philox sampler,
//                    = trs_batch_prepare); writes the per-triple score gradients gz+ / gz- (FM: through the sigmoid), the
//                    user-row gradient du (B,D), the loss sum, and an ownership mark per referenced row.
//   K2a item_update<1> re-reads the still-unmodified user rows (L2 / Infinity-Cache hot) and applies
//                    item[i] -= lr*gz+ * u, item[j] -= lr*gz- * u (+ 1-wide terms) for the references that OWN their
//                    row (their K1 mark survived: exactly one per distinct row) with plain whole-row read-modify-writes;
//                    stamps user rows that have several references (marks and stamps carry the step number: no resets).
//   K2b item_update<2> the other references of duplicated item rows add with float atomics (one 256-B wave-instruction
//                    per row segment).  K2a/K2b touch only item tables.
//   K3  user_update  user[u] -= lr*du (+ 1-wide term) from the staged rows: plain for rows referenced once, atomics for
//                    stamped rows.  Touches only user tables.
// Float atomics run at ~1.3 TB/s chip-wide against ~6 TB/s for plain stores (MI355X_MICROARCH.md "Global float
// atomics"), so every reference that is provably alone on its row takes the plain path.
#include "score_kernels.h"
#include "opt_rows.h"

#include <math.h>
#include <stdlib.h>

namespace trs {

struct FastArgs {
  trs_tables T;
  // batch ids (int32): outputs of K1 when `from_stream`, inputs otherwise
  int32_t* user;
  int32_t* pos;
  int32_t* neg;
  int64_t B;
  int32_t* err;
  // resident stream (from_stream != 0)
  int from_stream;
  const int2* sui;  // (N) interleaved {user, item}: one 8-byte random read per triple instead of two 4-byte ones
  const int32_t* neg_static;
  int64_t N;
  uint64_t shuffle_key;
  int hb;
  int64_t t0;
  uint64_t sample_seed;
  uint64_t sample_offset;
  // staging
  float* gz;  // (2,B): score gradients of the positive / negative pass (after the FM sigmoid)
  float* du;  // (B,D)
  float inv_B;
  float lr;
  int loss;  // TRS_LOSS_HINGE | TRS_LOSS_BPR
  float* loss_sum;
  // duplicate detection (all NULL: every update is atomic)
  uint64_t* uown;  // (n_users) last reference of the step that named the row: (stamp << 32) | t
  uint64_t* iown;  // (n_items) (stamp << 32) | (2t + which)
  uint32_t* udup;  // (n_users) == stamp: the row has more than one reference in this step
  uint32_t* idup;  // (n_items)
  uint32_t stamp;
  // presorted mode with precomputed per-position user-duplicate flags (trs_epoch_presort): K1 applies the user update
  // itself for users referenced once in the batch and stages the OLD user row for K2 instead of the gradient
  const uint8_t* udup_pos;  // (B) 1: this triple's user has another reference in the batch; NULL: mode off
  // (B,2) {pos, neg}: 1 = that item row has another reference in the batch; NULL: every item reference goes through the
  // sorted runs.  Given: K1 applies the item update of a reference that is alone on its row itself (the user row is in
  // registers and nobody else reads or writes that item row this step) and K2 only walks rows with several references
  const uint8_t* idup_pos;
  float* ustage;            // (B,D) pre-update user rows, read by the sorted item update
  OptArgs o;                // update rule of the presorted mode (kind OPT_SGD: lr above)
  // INL 3 (flag mode, one launch per step): sync[0] = ONE monotonic arrival counter (never reset, wraps mod 2^32); a
  // launch counts its workgroups in and waits until the counter has reached sync_target = arrivals of all earlier
  // launches + its own grid (wrap-safe signed compare); sync[32 * (1 + g)], g < 8: flag lines that carry the last
  // completed target (TRS_SYNC_WORDS uint32 in all)
  uint32_t* sync;
  uint32_t sync_base, sync_target;
  // flagged-first order (trs_epoch_flags_ordered): *nflag = the batch's triples that carry a flagged reference, all of them
  // at its first positions; NULL = any order
  const int32_t* nflag;
};

struct RawIds {   // loads issued, nothing consumed yet
  int32_t u, p, n;
  uint8_t dup;
  uint16_t idup;  // INL 2: {pos, neg} item-duplicate flags, one byte each
  int64_t v;      // SRC 1: uniform draw over n_items-1 values (the negative is v + (v >= pos))
  bool valid;
};

struct TripleIds {
  int32_t u, p, n;
  bool valid, ok, dup, pdup, ndup;
};

// SRC: 0 = ids given in user/pos/neg; 1 = resident stream + dynamic sampler; 2 = resident stream + static negatives.
// A template parameter, not a run-time branch: a conditional block that contains a load ends in s_waitcnt vmcnt(0),
// which would also drain the row gathers in flight.  issue_ids only ISSUES the loads (and does the id-independent
// Philox arithmetic); finalize_ids consumes them one iteration later, so the wait it implies covers loads that are
// older than every row gather still in flight (vmcnt is in-order).
template <int SRC, int INL>
__device__ __forceinline__ RawIds issue_ids(const FastArgs& a, int64_t t) {
  RawIds r;
  r.valid = t < a.B;
  const int64_t tc = r.valid ? t : a.B - 1;  // loads stay unconditional
  r.v = 0;
  if (SRC != 0) {
    const int64_t p = trs_feistel_perm(a.t0 + tc, a.N, a.shuffle_key, a.hb);
    const int2 ui = a.sui[p];
    r.u = ui.x;
    r.p = ui.y;
    if (SRC == 2) {
      r.n = a.neg_static[p];
    } else {
      const trs_u4 x = trs_philox4x32_10(a.sample_offset + (uint64_t)tc, a.sample_seed);
      const uint64_t x64 = ((uint64_t)x.y << 32) | (uint64_t)x.x;
      r.v = a.T.n_items > 1 ? (int64_t)trs_mulhi64(x64, (uint64_t)(a.T.n_items - 1)) : 0;
      r.n = 0;
    }
  } else {
    r.u = a.user[tc];
    r.p = a.pos[tc];
    r.n = a.neg[tc];
  }
  r.dup = 1;
  r.idup = 0x0101;
  if (INL) r.dup = a.udup_pos[tc];
  if (INL >= 2) r.idup = reinterpret_cast<const uint16_t*>(a.idup_pos)[tc];
  return r;
}

template <int SRC>
__device__ __forceinline__ TripleIds finalize_ids(const FastArgs& a, const RawIds& w) {
  TripleIds r;
  r.valid = w.valid;
  int64_t uid = w.u, pid = w.p;
  int64_t nid = SRC == 1 ? (a.T.n_items > 1 ? w.v + (w.v >= pid ? 1 : 0) : 0) : (int64_t)w.n;
  r.ok = true;
  if ((uint64_t)uid >= (uint64_t)a.T.n_users) { r.ok = false; uid = 0; }
  if ((uint64_t)pid >= (uint64_t)a.T.n_items) { r.ok = false; pid = 0; }
  if ((uint64_t)nid >= (uint64_t)a.T.n_items) { r.ok = false; nid = 0; }
  r.u = (int32_t)uid;
  r.p = (int32_t)pid;
  r.n = (int32_t)nid;
  r.dup = w.dup != 0;
  r.pdup = (w.idup & 0x00ffu) != 0;
  r.ndup = (w.idup & 0xff00u) != 0;
  return r;
}

template <int VEC, int K>
struct TripleRows {
  RowReg<VEC, K> u, pi, ni;
  float ul, pl, nl;
  RowReg<VEC, K> us1, us2;  // adaptive rules: the user row's optimiser state (exp_avg | sum, exp_avg_sq)
  float uls1, uls2;
};

template <int VEC, int G, int K, bool FULL, int OPT, int NTU = 0>
__device__ __forceinline__ void load_rows(TripleRows<VEC, K>& r, const trs_tables& T, const OptArgs& o,
                                          const TripleIds& id, int lig) {
  row_load<VEC, G, K, FULL, (NTU & 1) != 0>(r.u, T.user, id.u, T.D, lig);
  row_load<VEC, G, K, FULL>(r.pi, T.item, id.p, T.D, lig);
  row_load<VEC, G, K, FULL>(r.ni, T.item, id.n, T.D, lig);
  r.ul = T.user_lin[id.u];
  r.pl = T.item_lin[id.p];
  r.nl = T.item_lin[id.n];
  if (OPT != OPT_SGD) {  // compile-time: issued with the rows, unconditionally (duplicated users waste them: 6 % at c2)
    row_load<VEC, G, K, FULL>(r.us1, o.user_s1, id.u, T.D, lig);
    r.uls1 = o.user_lin_s1[id.u];
    if (OPT == OPT_ADAM) {
      row_load<VEC, G, K, FULL>(r.us2, o.user_s2, id.u, T.D, lig);
      r.uls2 = o.user_lin_s2[id.u];
    }
  }
}

// INL (presorted mode with user-duplicate flags): the user update of a triple whose user is referenced once in the
// batch is applied right here (the row is in registers and nobody else reads it this step); the OLD user row is staged
// for K2 instead of the gradient; only duplicated users stage their gradient for the small atomic pass K3'.
// OPT != OPT_SGD (INL only): the same in-place user update with the SparseAdam / Adagrad rule; its state rows travel with
// the user row.
// INL 2 (plain SGD, item-duplicate flags): an item reference that is alone on its row in the batch is updated right here
// too — item[i] += (-lr*gz) * u with the same two roundings as the sorted run of length one it replaces — and the old
// user row is staged only when one of the triple's item references still goes through the runs.  At c4 (65 536
// references over 1M items per step) 94 % of the references are alone: the step becomes one read and one write per row.
// INL 3 (flag mode as ONE launch per step; launch_fwd_stage picks it when every workgroup of the grid is resident at
// once): INL 2, and the flagged references are applied by this same launch instead of flagged_update_kernel.  The
// workgroup lists its flagged references in LDS — (triple, which) | table row | coefficient | coefficient of the 1-wide
// term; a wave takes its slots with ONE LDS atomic per iteration, positions inside come from ballots — and stages their
// rows in global memory as in INL 2.  Batches in arbitrary order (a.nflag NULL): after its last triple a
// workgroup waits for its loads and stores (s_waitcnt vmcnt(0) + barrier), counts itself in on the step's arrival
// counter and waits until ALL workgroups of the launch have done so: from then on no row of the tables is read any more
// by this step, so the float atomics of the flagged references — each workgroup re-reads the rows IT staged (same CU:
// its own stores are visible to it) — may land.  Flagged-first batches (a.nflag, the product path): the workgroups count
// themselves in right behind the flagged triples, mid-loop, and every wave applies its own references after its last
// iteration without waiting for anybody (see `early` below).  What this replaces: a kernel boundary, a ramp, and the
// second launch's detection pass over all 3B ids and flags for the 6 % of references that are flagged.  Every wait is
// bounded (50 ms of s_memrealtime): a grid that cannot become resident (another process on the same GPU) raises err bit
// 2 instead of hanging.
constexpr int DEFER_ITERS = 8;  // iterations per lane group the per-wave list is sized for (launch_fwd_stage keeps to it)
struct DeferEntry {
  uint32_t tw;   // (t << 2) | which      which: 0 user (row staged in du), 1 positive, 2 negative (row staged in ustage)
  int32_t row;   // table row
  float c;       // coefficient of the staged row
  float clin;    // increment of the row's 1-wide term
};

// NTU (INL 3): bit 0 = the user row is loaded nontemporally, bit 1 = the updated user row is stored nontemporally — user
// tables far beyond the Infinity Cache (c4: 5.1 GB) are read and written once per epoch per row; keeping them out leaves
// the cache to the item table, whose rows do come back (launch_fwd_stage decides by table size).
template <int NET, int VEC, int G, int K, int SRC, bool FULL, int INL, int OPT = OPT_SGD, int NTU = 0>
__global__ __launch_bounds__(TRS_BLOCK) void fwd_stage_kernel(const FastArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  constexpr bool DEFER = INL == 3;
  constexpr int NWV = TRS_BLOCK / TRS_WAVE;
  constexpr int LIST_CAP = DEFER ? NWV * DEFER_ITERS * TPW * 3 : 1;  // every reference of the workgroup's triples
  __shared__ DeferEntry s_list[LIST_CAP];  // ONE list per workgroup: slots handed out by an LDS counter
  __shared__ int s_total;                  // entries of s_list
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t stride = nwave * TPW;  // triples between two iterations of one lane group
  float loss_acc = 0.f;

#ifdef TRS_K1_STAMPS  // diagnostic build only (tools/k1_stamps.py): s_memrealtime (100 MHz) at four points of every workgroup
  uint64_t* stamps = reinterpret_cast<uint64_t*>(a.gz) + 4 * (int64_t)blockIdx.x;  // (gz is unused in INL 3)
  if (INL == 3 && threadIdx.x == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
#endif
  // Flagged-first batches (a.nflag): every flagged row of the step is read in the launch's first `fi` iterations, so a
  // workgroup counts itself in on the arrival counter as soon as its four waves are past them — mid-loop, with a plain
  // (non-returning) atomic nobody waits for — and at the end of its own loop only LOOKS at the counter: every other
  // workgroup counted itself in ~10 us ago, so the look finds the grid complete and the flagged references' atomics go
  // out at once.  No workgroup waits for the slowest one any more (time stamps before: main loops end at 15 / 18 / 23
  // us, everybody then waited until 24.5 and applied until 27-29).
  __shared__ int s_arr;   // waves of this workgroup that are past the flagged iterations
  __shared__ int s_left;  // early mode: some wave has not applied all its flagged references inside its loop
  int64_t fi = 0;
  bool early = false, arrived = false;
  if (DEFER) {
    if (threadIdx.x == 0) s_total = s_arr = s_left = 0;
    if (a.nflag) {
      const int64_t nf = *a.nflag;
      fi = (nf + stride - 1) / stride;
      early = fi < (B + stride - 1) / stride;  // (not: a batch left in its order reports nf = B)
    }
    __syncthreads();
  }
  auto arrive = [&]() {  // wave-uniform
    int before = 0;
    if (lane == 0) before = atomicAdd(&s_arr, 1);
    before = __builtin_amdgcn_readfirstlane(before);
    if (before == NWV - 1 && lane == 0)  // the workgroup's last wave: count the workgroup in (result unused: no wait)
      (void)__hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = true;
  };
  // early mode: a wave lists its flagged references in ITS OWN part of the list (no LDS atomic) and applies them itself
  // right after its last iteration — a wave's loads of the rows it staged come back behind its own stores to the same
  // addresses, so it needs no barrier with the other waves: the look at the arrival counter and the first round of staged
  // rows are issued behind the last row loads and come back with them.  (Applying them INSIDE the loop, a round per
  // iteration, was built: the 1 200 waves with something to apply looked at the counter's line once per iteration while
  // the arrivals were still landing on it — 43 against 33 us per step — and the extra live registers cost a wave per SIMD.)
  constexpr int CAPW = DEFER_ITERS * TPW * 3, DU = 4, KDD = (N * G + TRS_WAVE - 1) / TRS_WAVE;
  const int wv = threadIdx.x >> 6;
  int my_cnt = 0, next_e = 0;
  bool complete = false;
  auto entry_at = [&](int e) -> int { return early ? wv * CAPW + e : wv + NWV * e; };
  auto load_round = [&](float (&x)[DU][KDD], int e0, int n_mine) {
#pragma unroll
    for (int k = 0; k < DU; ++k) {
      const bool has = e0 + k < n_mine;
      // (same address in every lane: one LDS broadcast; without an entry, slot 0 is read as a dummy and not used)
      const DeferEntry en = s_list[has ? entry_at(e0 + k) : 0];
      const int64_t tk = has ? (int64_t)(en.tw >> 2) : 0;
      const float* src = ((en.tw & 3u) == 0 ? a.du : a.ustage) + tk * (int64_t)D;
#pragma unroll
      for (int q = 0; q < KDD; ++q) {
        const int e = q * TRS_WAVE + lane;
        x[k][q] = __hip_atomic_load(src + (e < D ? e : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past L1
      }
    }
  };
  auto apply_round = [&](float (&x)[DU][KDD], int e0, int n_mine) {
#pragma unroll
    for (int k = 0; k < DU; ++k) {
      if (e0 + k >= n_mine) continue;  // (wave-uniform)
      const DeferEntry en = s_list[entry_at(e0 + k)];
      const int which = (int)(en.tw & 3u);
      float* dst = (which == 0 ? T.user : T.item) + (int64_t)en.row * D;
#pragma unroll
      for (int q = 0; q < KDD; ++q) {
        const int e = q * TRS_WAVE + lane;
        if (e < D) atomicAdd(dst + e, en.c * x[k][q]);
      }
      if (lane == 0) atomicAdd((which == 0 ? T.user_lin : T.item_lin) + en.row, en.clin);
    }
  };
  int64_t t = wave * TPW + lane / G;
  // wave-uniform trip count: the wave's first group decides (its t is the smallest of the wave)
  const int64_t t_first = wave * TPW;
  const int64_t niter = t_first < B ? (B - t_first + stride - 1) / stride : 0;

  // Reduce one triple whose rows are in registers and write its staging (stores only: no waits on loads in flight).
  auto reduce = [&](TripleRows<VEC, K>& r, const TripleIds& id, int64_t tt) {
    const bool live = id.valid && id.ok;
    if (id.valid && !id.ok && lig == 0 && a.err) atomicOr(a.err, 1);
    float pp = 0.f, pn = 0.f;
    if (NET == TRS_NET_FM) {
      // (u+i)^2 - (u^2 + i^2) per element, exactly the reference's power_of_sum - sum_of_power (fm.py:83-86)
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float sp_ = r.u.v[n] + r.pi.v[n], sn_ = r.u.v[n] + r.ni.v[n];
        const float uu = r.u.v[n] * r.u.v[n];
        pp += sp_ * sp_ - (uu + r.pi.v[n] * r.pi.v[n]);
        pn += sn_ * sn_ - (uu + r.ni.v[n] * r.ni.v[n]);
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        pp += r.u.v[n] * r.pi.v[n];
        pn += r.u.v[n] * r.ni.v[n];
      }
    }
    pp = trs_group_sum<G>(pp);
    pn = trs_group_sum<G>(pn);
    float sp, sn;
    if (NET == TRS_NET_FM) {
      sp = sigmoidf_((r.ul + r.pl) + 0.5f * pp);
      sn = sigmoidf_((r.ul + r.nl) + 0.5f * pn);
    } else {
      sp = (pp + r.ul) + r.pl;
      sn = (pn + r.ul) + r.nl;
    }
    float lval, dneg;
    trs_pair_loss(a.loss, sp, sn, lval, dneg);
    const float act = live ? dneg : 0.f;
    float gp = -act * a.inv_B, gn = act * a.inv_B;
    if (live && lig == 0) loss_acc += lval;
    if (NET == TRS_NET_FM) {
      gp = gp * ((1.0f - sp) * sp);
      gn = gn * ((1.0f - sn) * sn);
    }
    if (DEFER) {  // list the flagged references of the wave's triples (one lane per group speaks for its triple); outside
      // every divergent branch: the ballots and the wave-uniform counter need all lanes
      const bool lead = lig == 0 && live;
      const uint64_t mu = __ballot(lead && id.dup), mp = __ballot(lead && id.pdup), mn = __ballot(lead && id.ndup);
      const uint64_t below = ((uint64_t)1 << lane) - 1;
      const int cu = __popcll(mu), cp = __popcll(mp), cn = __popcll(mn);
      int bu = 0;
      if (early) {  // the wave's own part of the list
        bu = wv * CAPW + my_cnt;
        my_cnt += cu + cp + cn;
        // a flagged reference BEHIND the first nf triples: n_flagged_dev does not describe these id / flag arrays (they
        // are not the ones trs_epoch_flags_ordered wrote) — the workgroup has counted itself in already, so this step is
        // not exact: err bit 3
        if (arrived && (cu + cp + cn) && lane == 0 && a.err) atomicOr(a.err, 8);
      } else if (cu + cp + cn) {  // (wave-uniform) this wave's slots of the workgroup's list: one LDS atomic per iteration
        if (lane == 0) bu = atomicAdd(&s_total, cu + cp + cn);
        bu = __builtin_amdgcn_readfirstlane(bu);
      }
      DeferEntry* L = s_list;
      const int bp = bu + cu, bn = bp + cp;
      if (lead && id.dup) L[bu + __popcll(mu & below)] = {(uint32_t)tt << 2, id.u, -a.lr, -a.lr * (gp + gn)};
      if (lead && id.pdup) L[bp + __popcll(mp & below)] = {((uint32_t)tt << 2) | 1u, id.p, -a.lr * gp, -a.lr * gp};
      if (lead && id.ndup) L[bn + __popcll(mn & below)] = {((uint32_t)tt << 2) | 2u, id.n, -a.lr * gn, -a.lr * gn};
    }
    if (id.valid) {
      RowReg<VEC, K> g;
#pragma unroll
      for (int n = 0; n < N; ++n) g.v[n] = gp * r.pi.v[n] + gn * r.ni.v[n];
      if (INL) {
        if (INL < 2 || id.pdup || id.ndup) row_store<VEC, G, K>(r.u, a.ustage + tt * (int64_t)D, D, lig);
        if (INL >= 2 && live) {
          const float cp = -a.lr * gp, cn = -a.lr * gn;  // the coefficient the sorted run forms from gz
          if (!id.pdup) {
            RowReg<VEC, K> o;
#pragma unroll
            for (int n = 0; n < N; ++n) o.v[n] = r.pi.v[n] + cp * r.u.v[n];
            row_store<VEC, G, K>(o, T.item + id.p * (int64_t)D, D, lig);
            if (lig == 0) T.item_lin[id.p] = r.pl + cp;
          }
          if (!id.ndup) {
            RowReg<VEC, K> o;
#pragma unroll
            for (int n = 0; n < N; ++n) o.v[n] = r.ni.v[n] + cn * r.u.v[n];
            row_store<VEC, G, K>(o, T.item + id.n * (int64_t)D, D, lig);
            if (lig == 0) T.item_lin[id.n] = r.nl + cn;
          }
        }
        if (id.dup || !live) {
          row_store<VEC, G, K>(g, a.du + tt * (int64_t)D, D, lig);
        } else {
          RowReg<VEC, K> un;
          if (OPT == OPT_SGD) {
#pragma unroll
            for (int n = 0; n < N; ++n) un.v[n] = r.u.v[n] + (-a.lr) * g.v[n];
            row_store<VEC, G, K, (NTU & 2) != 0>(un, T.user + id.u * (int64_t)D, D, lig);
            if (lig == 0) T.user_lin[id.u] = r.ul + (-a.lr) * (gp + gn);
          } else {
#pragma unroll
            for (int n = 0; n < N; ++n) un.v[n] = opt_apply<OPT>(r.u.v[n], g.v[n], r.us1.v[n], r.us2.v[n], a.o);
            row_store<VEC, G, K>(un, T.user + id.u * (int64_t)D, D, lig);
            row_store<VEC, G, K>(r.us1, a.o.user_s1 + id.u * (int64_t)D, D, lig);
            if (OPT == OPT_ADAM) row_store<VEC, G, K>(r.us2, a.o.user_s2 + id.u * (int64_t)D, D, lig);
            if (lig == 0) {
              T.user_lin[id.u] = opt_apply<OPT>(r.ul, gp + gn, r.uls1, r.uls2, a.o);
              a.o.user_lin_s1[id.u] = r.uls1;
              if (OPT == OPT_ADAM) a.o.user_lin_s2[id.u] = r.uls2;
            }
          }
        }
      } else {
        row_store<VEC, G, K>(g, a.du + tt * (int64_t)D, D, lig);
      }
      if (lig == 0) {
        if (!DEFER) {  // (INL 3: the coefficients travel in the list)
          a.gz[tt] = gp;
          a.gz[B + tt] = gn;
        }
        if (SRC != 0) {  // the batch this step was derived from, for K1b / K2 / K3 (and for the caller)
          a.user[tt] = id.u;
          a.pos[tt] = id.p;
          a.neg[tt] = id.n;
        }
        if (!INL && live && a.uown) {
          const uint64_t hi = (uint64_t)a.stamp << 32;
          a.uown[id.u] = hi | (uint64_t)(uint32_t)tt;
          if (a.iown) {  // wave-uniform; not needed when the item references were presorted (presort.hip)
            a.iown[id.p] = hi | (uint64_t)(uint32_t)(2 * tt);
            a.iown[id.n] = hi | (uint64_t)(uint32_t)(2 * tt + 1);
          }
        }
      }
    }
  };

  // Software pipeline, unrolled by two so that nothing in flight is ever copied between registers (a copy of a loaded
  // value forces the wait for it): triple k's rows live in the "even" or "odd" register set by parity; in triple k's
  // phase the ids of k+2 are issued, the ids of k+1 consumed and its rows issued, then triple k is reduced.  Program
  // order = issue order, so each wait covers only loads older than everything that should stay in flight.
  TripleIds idE, idO;
  TripleRows<VEC, K> rE, rO;
  {
  // (INL 3: handing the workgroup's (iteration, wave) chunks out dynamically through an LDS counter was built and
  // measured: no change — the launch's slow tail is whole workgroups on slower XCDs / CUs, not single waves)
  const int64_t niter2 = (niter + 1) & ~(int64_t)1;  // even trip count: surplus phases run on clamped, invalid triples
  RawIds wE = issue_ids<SRC, INL>(a, t);
  RawIds wO = issue_ids<SRC, INL>(a, t + stride);
  idE = finalize_ids<SRC>(a, wE);
  load_rows<VEC, G, K, FULL, OPT, NTU>(rE, T, a.o, idE, lig);
  if (DEFER && early && fi == 0) arrive();
  for (int64_t it = 0; it < niter2; it += 2) {
    wE = issue_ids<SRC, INL>(a, t + 2 * stride);
    idO = finalize_ids<SRC>(a, wO);
    load_rows<VEC, G, K, FULL, OPT, NTU>(rO, T, a.o, idO, lig);
    reduce(rE, idE, t);
    if (DEFER && early && it + 1 == fi) arrive();  // (the rows already in flight belong to triples without a flagged reference)

    wO = issue_ids<SRC, INL>(a, t + 3 * stride);
    idE = finalize_ids<SRC>(a, wE);
    load_rows<VEC, G, K, FULL, OPT, NTU>(rE, T, a.o, idE, lig);
    reduce(rO, idO, t + stride);
    if (DEFER && early && it + 2 == fi) arrive();
    t += 2 * stride;
  }
  if (DEFER && early && !arrived) arrive();  // a wave with fewer iterations than fi (the batch's tail)
  }
#ifdef TRS_K1_STAMPS
  if (INL == 3 && threadIdx.x == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ float s_loss[TRS_BLOCK / TRS_WAVE];
  const float wl = trs_wave_sum(loss_acc);
  if (lane == 0) s_loss[threadIdx.x >> 6] = wl;
  if (DEFER && early && my_cnt > 0) {  // (wave-uniform)
    float x[DU][KDD];
    const uint32_t seen = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    load_round(x, 0, my_cnt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    complete = (int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)seen) - a.sync_target) >= 0;
    if (complete) {  // every workgroup of the launch is past its flagged triples: nobody reads these rows any more
      apply_round(x, 0, my_cnt);
      for (int e0 = DU; e0 < my_cnt; e0 += DU) {
        load_round(x, e0, my_cnt);
        apply_round(x, e0, my_cnt);
      }
      next_e = my_cnt;
    } else if (lane == 0) {
      s_left = 1;  // (not yet: the workgroup waits below)
    }
  }
  if (DEFER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's row loads and staging stores are done
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < TRS_BLOCK / TRS_WAVE; ++w) L += s_loss[w];
    if (L != 0.f) atomicAdd(a.loss_sum, L);
  }
  if (DEFER) {
    // Not early: the workgroup's flagged references are dealt round-robin to its four waves (a single wave's share of
    // them is 0 to 6 at c4, and the longest one of the launch would set the kernel's end), DU at a time: the staged rows
    // of a round are loaded together (L2: written by this CU a moment ago), then one float atomic per element, a row =
    // adjacent dwords (the full-rate atomic shape).  Early: what a wave has not applied inside its loop (more than DU
    // references per iteration left, or a grid that was not counted in yet) — usually nothing, and then the workgroup
    // does not even look at the counter.
    const int total = s_total;
    const int n_mine = early ? my_cnt : (total > wv ? (total - wv + NWV - 1) / NWV : 0);
    const int e_first = early ? next_e : 0;
#ifdef TRS_K1_STAMPS
    if (early && !s_left && threadIdx.x == 0) stamps[2] = stamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
    if (early && !s_left) return;  // (workgroup-uniform: written before the barrier above)
    float x[DU][KDD];
    // the first round's staged rows travel to registers WHILE the workgroup waits for the grid (they are this
    // workgroup's own data, complete since the barrier above; typical: 2-3 flagged references per wave = one round)
    load_round(x, e_first, n_mine);
    if (threadIdx.x == 0 && early) {
      // counted in long ago, and so has everybody else: one look at the counter (bounded wait as below if it is not so)
      const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
      const uint64_t limit = (a.err && (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) ? 0ull : 5000000ull;
      while ((int32_t)(__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.sync_target) < 0) {
        __builtin_amdgcn_s_sleep(6);
        if (__builtin_amdgcn_s_memrealtime() - t_start > limit) {
          if (a.err) atomicOr(a.err, 4);
          break;
        }
      }
    } else if (threadIdx.x == 0) {
      // Arrivals are returning atomics on the counter's line: the workgroup whose add completes the grid publishes the
      // target on eight flag lines (128 B apart); everybody else polls only its flag line (blockIdx % 8), which nobody
      // writes meanwhile.  (All 512 workgroups polling the counter itself queued their reads between the arrivals on one
      // memory channel: + 30 us per launch; one designated poller on the counter + flags: 33.4 us per launch.)
      uint32_t* ctr = a.sync;
      const uint32_t before = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int32_t)(before + 1u - a.sync_target) >= 0) {  // the last workgroup of the launch: publish
#pragma unroll
        for (int g = 0; g < 8; ++g)
          __hip_atomic_store(a.sync + 32 * (1 + g), a.sync_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const uint64_t t_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        uint32_t* line = a.sync + 32 * (1 + (blockIdx.x & 7u));
        // bounded wait: 50 ms — and none at all once a launch has timed out (err bit 2 is sticky until the host reads it:
        // a run whose grids cannot be resident fails at its next error check, it does not spin 50 ms per step)
        const uint64_t limit = (a.err && (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) ? 0ull : 5000000ull;
        while ((int32_t)(__hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.sync_target) < 0) {
          __builtin_amdgcn_s_sleep(6);
          if (__builtin_amdgcn_s_memrealtime() - t_start > limit) {  // the grid is not resident at once
            if (a.err) atomicOr(a.err, 4);
            break;
          }
        }
      }
    }
    __syncthreads();
#ifdef TRS_K1_STAMPS
    if (threadIdx.x == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
    // every row read of the step is behind us, chip-wide
    apply_round(x, e_first, n_mine);
    for (int e0 = e_first + DU; e0 < n_mine; e0 += DU) {
      load_round(x, e0, n_mine);
      apply_round(x, e0, n_mine);
    }
#ifdef TRS_K1_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
#endif
  }
}

// Flag mode (the sparse regime: duplicate flags from trs_epoch_flags, no sorted runs), second launch of a step.  K1 has
// updated every row whose reference is alone in the batch and staged, for the flagged references, what their update
// needs: gz, the old user row (ustage) for item references, the gradient row (du) for users.  All reads of the step
// happened in K1, so the flagged references now add their contributions straight into the tables with float atomics:
//   item[row] += (-lr * gz) * ustage[t]      user[u] += -lr * du[t]      (+ the 1-wide terms)
// One lane per reference of the batch for detection (3B: users, positives, negatives; FLG_REFS per wave, ids and flags
// read coalesced), then the wave walks its flagged references FLG_U at a time: the staged rows of a round are loaded
// together, every lane one element per instruction — a row is adjacent dwords, the full-rate atomic shape
// (MI355X_MICROARCH.md "Global float atomics").  c4: 6 % of the item references, 3 % of the users (hashed flags).
constexpr int FLG_REFS = 32, FLG_U = 4;

template <int KD>  // ceil(D / 64)
__global__ __launch_bounds__(TRS_BLOCK) void flagged_update_kernel(const FastArgs a) {
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B, n = 3 * a.B;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  for (int64_t base = wave * FLG_REFS; base < n; base += nwave * FLG_REFS) {
    const int64_t r = base + lane;
    const bool valid = lane < FLG_REFS && r < n;
    const int64_t rc = valid ? r : base;
    const int which = (int)(rc / B);  // 0 user, 1 positive, 2 negative
    const int t = (int)(rc - (int64_t)which * B);
    const int32_t row = which == 0 ? a.user[t] : (which == 1 ? a.pos[t] : a.neg[t]);
    const uint8_t flag = which == 0 ? a.udup_pos[t] : a.idup_pos[2 * t + (which - 1)];
    // coefficient of the staged row: -lr for a user's gradient row, -lr*gz of the pass for an item reference
    const float gp = a.gz[t], gn = a.gz[B + t];
    const float c = which == 0 ? -a.lr : (-a.lr * (which == 1 ? gp : gn));
    const float clin = which == 0 ? -a.lr * (gp + gn) : c;
    uint64_t mask = __ballot(valid && flag != 0);
    while (mask) {
      int l[FLG_U];
      bool has[FLG_U];
      float x[FLG_U][KD], cc[FLG_U];
      float* dst[FLG_U];
#pragma unroll
      for (int k = 0; k < FLG_U; ++k) {
        has[k] = mask != 0;
        l[k] = has[k] ? __ffsll((unsigned long long)mask) - 1 : 0;
        mask &= mask - 1;  // (0 & anything stays 0)
        const int wk = __shfl(which, l[k], 64);
        const int64_t tk = __shfl(t, l[k], 64), rk = __shfl(row, l[k], 64);
        cc[k] = __shfl(c, l[k], 64);
        const float* src = (wk == 0 ? a.du : a.ustage) + tk * (int64_t)D;
        dst[k] = (wk == 0 ? T.user : T.item) + rk * (int64_t)D;
#pragma unroll
        for (int q = 0; q < KD; ++q) {
          const int e = q * TRS_WAVE + lane;
          x[k][q] = src[e < D ? e : 0];  // unconditional (clamped) loads: all FLG_U rows in flight together
        }
        const float cl = __shfl(clin, l[k], 64);  // (every lane shuffles: a shuffle under `lane == 0` reads a dead lane)
        if (lane == 0 && has[k]) atomicAdd((wk == 0 ? T.user_lin : T.item_lin) + rk, cl);
      }
#pragma unroll
      for (int k = 0; k < FLG_U; ++k) {
        if (!has[k]) continue;
#pragma unroll
        for (int q = 0; q < KD; ++q) {
          const int e = q * TRS_WAVE + lane;
          if (e < D) atomicAdd(dst[k] + e, cc[k] * x[k][q]);
        }
      }
    }
  }
}

struct UpdIds {
  int64_t u, p, n;
  bool live;
};

__device__ __forceinline__ UpdIds upd_ids(const FastArgs& a, int64_t t) {
  UpdIds r;
  const bool valid = t < a.B;
  const int64_t tc = valid ? t : a.B - 1;
  r.u = a.user[tc];
  r.p = a.pos[tc];
  r.n = a.neg[tc];
  const bool ok = (uint64_t)r.u < (uint64_t)a.T.n_users && (uint64_t)r.p < (uint64_t)a.T.n_items &&
                  (uint64_t)r.n < (uint64_t)a.T.n_items;  // bad ids were flagged by K1
  if (!ok) r.u = r.p = r.n = 0;
  r.live = valid && ok;
  return r;
}

// ---- plain (non-atomic) update passes in the K1 lane mapping: G lanes x VEC floats per row, 16 B per lane ------------
// Rows whose update needs no atomic (one owner per distinct item row; user rows referenced once) are read, modified and
// written whole with 16-byte accesses.  A reference that does not qualify loads row 0 instead (a hot line) and stores
// nothing: no conditional block contains a load, and no random row is fetched for nothing.
template <int VEC, int G, int K, bool FULL>
__global__ __launch_bounds__(TRS_BLOCK) void item_owner_update_kernel(const FastArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  constexpr int U = 2;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t per_it = (int64_t)TPW * U;
  const int64_t niter = (B + per_it - 1) / per_it;
  const uint64_t hi = (uint64_t)a.stamp << 32;
  for (int64_t it = wave; it < niter; it += nwave) {
    UpdIds id[U];
    int64_t t[U];
    bool po[U], no[U], ud[U];
    float cp[U], cn[U], pl[U], nl[U];
    RowReg<VEC, K> u[U], pr[U], nr[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      t[k] = it * per_it + k * TPW + lane / G;
      id[k] = upd_ids(a, t[k]);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tc = id[k].live ? t[k] : 0;
      cp[k] = -a.lr * a.gz[tc];
      cn[k] = -a.lr * a.gz[B + tc];
      po[k] = id[k].live && a.iown[id[k].p] == (hi | (uint64_t)(uint32_t)(2 * tc));
      no[k] = id[k].live && a.iown[id[k].n] == (hi | (uint64_t)(uint32_t)(2 * tc + 1));
      ud[k] = id[k].live && a.uown[id[k].u] != (hi | (uint64_t)(uint32_t)tc);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t pr_ = po[k] ? id[k].p : 0, nr_ = no[k] ? id[k].n : 0;
      row_load<VEC, G, K, FULL>(u[k], T.user, (po[k] || no[k]) ? id[k].u : 0, D, lig);
      row_load<VEC, G, K, FULL>(pr[k], T.item, pr_, D, lig);
      row_load<VEC, G, K, FULL>(nr[k], T.item, nr_, D, lig);
      pl[k] = T.item_lin[pr_];
      nl[k] = T.item_lin[nr_];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (po[k]) {
        RowReg<VEC, K> o;
#pragma unroll
        for (int n = 0; n < N; ++n) o.v[n] = pr[k].v[n] + cp[k] * u[k].v[n];
        row_store<VEC, G, K>(o, T.item + id[k].p * (int64_t)D, D, lig);
        if (lig == 0) T.item_lin[id[k].p] = pl[k] + cp[k];
      }
      if (no[k]) {
        RowReg<VEC, K> o;
#pragma unroll
        for (int n = 0; n < N; ++n) o.v[n] = nr[k].v[n] + cn[k] * u[k].v[n];
        row_store<VEC, G, K>(o, T.item + id[k].n * (int64_t)D, D, lig);
        if (lig == 0) T.item_lin[id[k].n] = nl[k] + cn[k];
      }
      if (ud[k] && lig == 0) a.udup[id[k].u] = a.stamp;
    }
  }
}

template <int VEC, int G, int K, bool FULL>
__global__ __launch_bounds__(TRS_BLOCK) void user_plain_update_kernel(const FastArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  constexpr int U = 2;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t per_it = (int64_t)TPW * U;
  const int64_t niter = (B + per_it - 1) / per_it;
  for (int64_t it = wave; it < niter; it += nwave) {
    UpdIds id[U];
    int64_t t[U];
    bool al[U];
    float c[U], wl[U];
    RowReg<VEC, K> g[U], w[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      t[k] = it * per_it + k * TPW + lane / G;
      id[k] = upd_ids(a, t[k]);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tc = id[k].live ? t[k] : 0;
      c[k] = -a.lr * (a.gz[tc] + a.gz[B + tc]);
      al[k] = id[k].live && a.udup[id[k].u] != a.stamp;
      row_load<VEC, G, K, FULL>(g[k], a.du, tc, D, lig);
      row_load<VEC, G, K, FULL>(w[k], T.user, al[k] ? id[k].u : 0, D, lig);
      wl[k] = T.user_lin[al[k] ? id[k].u : 0];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (!id[k].live) continue;
      float* urow = T.user + id[k].u * (int64_t)D;
      if (al[k]) {
        RowReg<VEC, K> o;
#pragma unroll
        for (int n = 0; n < N; ++n) o.v[n] = w[k].v[n] + (-a.lr) * g[k].v[n];
        row_store<VEC, G, K>(o, urow, D, lig);
        if (lig == 0) T.user_lin[id[k].u] = wl[k] + c[k];
      } else {  // duplicated user row (rare): float atomics, element by element
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
          const int e = (kk * G + lig) * VEC;
#pragma unroll
          for (int q = 0; q < VEC; ++q)
            if (e + q < D) atomicAdd(urow + e + q, -a.lr * g[k].v[kk * VEC + q]);
        }
        if (lig == 0) atomicAdd(T.user_lin + id[k].u, c[k]);
      }
    }
  }
}

// K3' (presorted mode with user-duplicate flags): only triples whose user has other references in the batch still owe
// their user update; they add their staged gradient with float atomics (c2: 6 % of the triples).
template <int VEC, int G, int K, bool FULL>
__global__ __launch_bounds__(TRS_BLOCK) void user_dup_update_kernel(const FastArgs a) {
  constexpr int TPW = TRS_WAVE / G;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t niter = (B + TPW - 1) / TPW;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t t = it * TPW + lane / G;
    const UpdIds id = upd_ids(a, t);
    const int64_t tc = id.live ? t : 0;
    const bool dup = id.live && a.udup_pos[tc] != 0;
    const float c = -a.lr * (a.gz[tc] + a.gz[B + tc]);
    RowReg<VEC, K> g;
    row_load<VEC, G, K, FULL>(g, a.du, dup ? tc : 0, D, lig);
    if (!dup) continue;
    float* urow = T.user + id.u * (int64_t)D;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      const int e = (kk * G + lig) * VEC;
#pragma unroll
      for (int q = 0; q < VEC; ++q)
        if (e + q < D) atomicAdd(urow + e + q, -a.lr * g.v[kk * VEC + q]);
    }
    if (lig == 0) atomicAdd(T.user_lin + id.u, c);
  }
}

// K2 / K3 mapping: LPR lanes cover one row (4 B per lane: a D = 64 row is ONE 256-B wave-instruction, the full-rate
// float-atomic shape); a wave works on U x (64 / LPR) triples per iteration with every load of the iteration issued
// before the first store / atomic (no conditional block contains a load), KD = ceil(D / LPR) elements per lane.
// MODE 0: every reference updates its item row with float atomics (no scratch).
// MODE 1 (phase a): only the reference that OWNS its row (its K1 mark survived) updates it, with a plain whole-row
//         read-modify-write — one owner per distinct row, so no two writers; also stamps duplicated user rows for K3.
// MODE 2 (phase b, a later launch): the remaining references add with float atomics.
// Splitting by ownership instead of "row has duplicates" moves one reference of every duplicated row (c2: 56 % of all
// item references are owners) from the ~1.3 TB/s atomic path to plain stores.
template <int LPR, int KD, int U, int MODE>
__global__ __launch_bounds__(TRS_BLOCK) void item_update_kernel(const FastArgs a) {
  constexpr int EPW = TRS_WAVE / LPR;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lir = lane % LPR;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t per_it = (int64_t)EPW * U;
  const int64_t niter = (B + per_it - 1) / per_it;
  const uint64_t hi = (uint64_t)a.stamp << 32;
  for (int64_t it = wave; it < niter; it += nwave) {
    UpdIds id[U];
    float cp[U], cn[U], uv[U][KD], pv[U][KD], nv[U][KD], pl[U], nl[U];
    bool po[U], no[U], ud[U];  // pos / neg reference owns its row; user row is duplicated
    int64_t t[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      t[k] = it * per_it + k * EPW + lane / LPR;
      id[k] = upd_ids(a, t[k]);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tc = id[k].live ? t[k] : 0;
      cp[k] = -a.lr * a.gz[tc];
      cn[k] = -a.lr * a.gz[B + tc];
      po[k] = no[k] = ud[k] = false;
      if (MODE != 0) {
        po[k] = a.iown[id[k].p] == (hi | (uint64_t)(uint32_t)(2 * tc));
        no[k] = a.iown[id[k].n] == (hi | (uint64_t)(uint32_t)(2 * tc + 1));
      }
      if (MODE == 1) ud[k] = a.uown[id[k].u] != (hi | (uint64_t)(uint32_t)tc);
#pragma unroll
      for (int q = 0; q < KD; ++q) {
        const int d = q * LPR + lir;
        const int dc = d < D ? d : 0;
        uv[k][q] = T.user[((MODE == 2 && po[k] && no[k]) ? 0 : id[k].u) * (int64_t)D + dc];
        if (MODE == 1) {
          pv[k][q] = T.item[id[k].p * (int64_t)D + dc];
          nv[k][q] = T.item[id[k].n * (int64_t)D + dc];
        }
      }
      if (MODE == 1) {
        pl[k] = T.item_lin[id[k].p];
        nl[k] = T.item_lin[id[k].n];
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (!id[k].live) continue;
      float* prow = T.item + id[k].p * (int64_t)D;
      float* nrow = T.item + id[k].n * (int64_t)D;
#pragma unroll
      for (int q = 0; q < KD; ++q) {
        const int d = q * LPR + lir;
        if (d < D) {
          if (MODE == 1) {
            if (po[k]) prow[d] = pv[k][q] + cp[k] * uv[k][q];
            if (no[k]) nrow[d] = nv[k][q] + cn[k] * uv[k][q];
          } else {
            if (!po[k]) atomicAdd(prow + d, cp[k] * uv[k][q]);
            if (!no[k]) atomicAdd(nrow + d, cn[k] * uv[k][q]);
          }
        }
      }
      if (lir == 0) {
        if (MODE == 1) {
          if (po[k]) T.item_lin[id[k].p] = pl[k] + cp[k];
          if (no[k]) T.item_lin[id[k].n] = nl[k] + cn[k];
          if (ud[k]) a.udup[id[k].u] = a.stamp;
        } else {
          if (!po[k]) atomicAdd(T.item_lin + id[k].p, cp[k]);
          if (!no[k]) atomicAdd(T.item_lin + id[k].n, cn[k]);
        }
      }
    }
  }
}

template <int LPR, int KD, int U, bool DEDUP>
__global__ __launch_bounds__(TRS_BLOCK) void user_update_kernel(const FastArgs a) {
  constexpr int EPW = TRS_WAVE / LPR;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.B;
  const int lane = threadIdx.x & 63;
  const int lir = lane % LPR;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t per_it = (int64_t)EPW * U;
  const int64_t niter = (B + per_it - 1) / per_it;
  for (int64_t it = wave; it < niter; it += nwave) {
    UpdIds id[U];
    float c[U], gv[U][KD], wv[U][KD], wl[U];
    bool al[U];
    int64_t t[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      t[k] = it * per_it + k * EPW + lane / LPR;
      id[k] = upd_ids(a, t[k]);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tc = id[k].live ? t[k] : 0;
      c[k] = -a.lr * (a.gz[tc] + a.gz[B + tc]);
      al[k] = DEDUP ? a.udup[id[k].u] != a.stamp : false;
      wl[k] = T.user_lin[id[k].u];
#pragma unroll
      for (int q = 0; q < KD; ++q) {
        const int d = q * LPR + lir;
        const int dc = d < D ? d : 0;
        gv[k][q] = a.du[tc * (int64_t)D + dc];
        wv[k][q] = T.user[id[k].u * (int64_t)D + dc];
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (!id[k].live) continue;
      float* urow = T.user + id[k].u * (int64_t)D;
#pragma unroll
      for (int q = 0; q < KD; ++q) {
        const int d = q * LPR + lir;
        if (d < D) {
          if (al[k]) urow[d] = wv[k][q] + (-a.lr) * gv[k][q]; else atomicAdd(urow + d, -a.lr * gv[k][q]);
        }
      }
      if (lir == 0) {
        if (al[k]) T.user_lin[id[k].u] = wl[k] + c[k]; else atomicAdd(T.user_lin + id[k].u, c[k]);
      }
    }
  }
}

// K1 of a metadata scorer (MT = 1..4 columns, ids and metadata ids given by position): the arithmetic of
// score_kernel<MODE 2> (score_kernels.h) with every gather of a triple — user, two items, 2*MT metadata rows and their
// 1-wide terms — issued before anything is consumed (the generic scorer walks the columns in a loop and the two passes
// one after the other: three dependent rounds of latency per triple).
// The north-star pass by itself — fused positive + negative embedding gather and Linear / FM pairwise score, no metadata,
// int32 ids — in the software pipeline of fwd_stage_kernel (ids two triples ahead, the three row gathers of the next
// triple in flight while the current one is reduced).  Same arithmetic as score_kernel<MODE 0>; used by
// trs_score_forward (evaluate(), forward_pair) when the shape allows.
template <int NET, int VEC, int G, int K, bool FULL, int NTM = 0>
__global__ __launch_bounds__(TRS_BLOCK) void pair_scores_kernel(const ScoreArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.Bt.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t stride = nwave * TPW;
  const int32_t* user = (const int32_t*)a.Bt.user;
  const int32_t* pos = (const int32_t*)a.Bt.pos;
  const int32_t* neg = (const int32_t*)a.Bt.neg;
  struct Ids {
    int32_t u, p, n;
    bool valid, ok;
  };
  struct Rows {
    RowReg<VEC, K> u, pi, ni;
    float ul, pl, nl;
  };
  auto issue = [&](int64_t t) {
    Ids r;
    r.valid = t < B;
    const int64_t tc = r.valid ? t : B - 1;
    r.u = user[tc]; r.p = pos[tc]; r.n = neg[tc];
    r.ok = true;
    return r;
  };
  auto finalize = [&](Ids& r) {
    if ((uint32_t)r.u >= (uint64_t)T.n_users) { r.ok = false; r.u = 0; }
    if ((uint32_t)r.p >= (uint64_t)T.n_items) { r.ok = false; r.p = 0; }
    if ((uint32_t)r.n >= (uint64_t)T.n_items) { r.ok = false; r.n = 0; }
  };
  auto gather = [&](Rows& r, const Ids& id) {  // NTM (experiment): bit 0 = user rows nontemporal, bit 1 = item rows
    row_load<VEC, G, K, FULL, (NTM & 1) != 0>(r.u, T.user, id.u, D, lig);
    row_load<VEC, G, K, FULL, (NTM & 2) != 0>(r.pi, T.item, id.p, D, lig);
    row_load<VEC, G, K, FULL, (NTM & 2) != 0>(r.ni, T.item, id.n, D, lig);
    r.ul = T.user_lin[id.u]; r.pl = T.item_lin[id.p]; r.nl = T.item_lin[id.n];
  };
  auto reduce = [&](const Rows& r, const Ids& id, int64_t t) {
    if (id.valid && !id.ok && lig == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
    float pp = 0.f, pn = 0.f;
    if (NET == TRS_NET_FM) {  // (u+i)^2 - (u^2 + i^2) per element: the reference's power_of_sum - sum_of_power
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float sp_ = r.u.v[n] + r.pi.v[n], sn_ = r.u.v[n] + r.ni.v[n];
        const float uu = r.u.v[n] * r.u.v[n];
        pp += sp_ * sp_ - (uu + r.pi.v[n] * r.pi.v[n]);
        pn += sn_ * sn_ - (uu + r.ni.v[n] * r.ni.v[n]);
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        pp += r.u.v[n] * r.pi.v[n];
        pn += r.u.v[n] * r.ni.v[n];
      }
    }
    pp = trs_group_sum<G>(pp);
    pn = trs_group_sum<G>(pn);
    float sp, sn;
    if (NET == TRS_NET_FM) {
      sp = sigmoidf_((r.ul + r.pl) + 0.5f * pp);
      sn = sigmoidf_((r.ul + r.nl) + 0.5f * pn);
    } else {
      sp = (pp + r.ul) + r.pl;
      sn = (pn + r.ul) + r.nl;
    }
    if (id.valid && lig == 0) {
      const bool live = id.ok;
      a.pos_score[t] = live ? sp : 0.f;
      a.neg_score[t] = live ? sn : 0.f;
    }
  };
  int64_t t = wave * TPW + lane / G;
  const int64_t t_first = wave * TPW;
  const int64_t niter = t_first < B ? (B - t_first + stride - 1) / stride : 0;
  const int64_t niter2 = (niter + 1) & ~(int64_t)1;
  Ids wE = issue(t), wO = issue(t + stride), idE, idO;
  Rows rE, rO;
  idE = wE;
  finalize(idE);
  gather(rE, idE);
  for (int64_t it = 0; it < niter2; it += 2) {
    wE = issue(t + 2 * stride);
    idO = wO;
    finalize(idO);
    gather(rO, idO);
    reduce(rE, idE, t);

    wO = issue(t + 3 * stride);
    idE = wE;
    finalize(idE);
    gather(rE, idE);
    reduce(rO, idO, t + stride);
    t += 2 * stride;
  }
}

template <int MT>
struct MetaIds {  // loads issued (raw) or clamped (final)
  int32_t u, p, n, mp[MT], mn[MT];
  uint8_t dup;
  bool valid, ok;
};
template <int VEC, int K, int MT>
struct MetaRows {
  RowReg<VEC, K> u, pi, ni, rp[MT], rn[MT];
  float ul, pl, nl, lp[MT], ln[MT];
  RowReg<VEC, K> us1, us2;  // adaptive rules: the user row's optimiser state, as in fwd_stage_kernel
  float uls1, uls2;
};

template <int NET, int VEC, int G, int K, bool FULL, int MT, int OPT = OPT_SGD>
__global__ __launch_bounds__(TRS_BLOCK) void meta_stage_kernel(const ScoreArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t B = a.Bt.B;
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const int64_t stride = nwave * TPW;
  const int32_t* user = (const int32_t*)a.Bt.user;
  const int32_t* pos = (const int32_t*)a.Bt.pos;
  const int32_t* neg = (const int32_t*)a.Bt.neg;
  const int32_t* pmeta = (const int32_t*)a.Bt.pos_meta;
  const int32_t* nmeta = (const int32_t*)a.Bt.neg_meta;
  float loss_acc = 0.f;

  auto issue = [&](int64_t t) {  // only issues the id loads
    MetaIds<MT> r;
    r.valid = t < B;
    const int64_t tc = r.valid ? t : B - 1;
    r.u = user[tc]; r.p = pos[tc]; r.n = neg[tc];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      r.mp[m] = pmeta[tc * MT + m];
      r.mn[m] = nmeta[tc * MT + m];
    }
    r.dup = a.udup_pos[tc];
    r.ok = true;
    return r;
  };
  auto finalize = [&](MetaIds<MT>& r) {  // consumes them: range checks, clamped ids
    if ((uint32_t)r.u >= (uint64_t)T.n_users) { r.ok = false; r.u = 0; }
    if ((uint32_t)r.p >= (uint64_t)T.n_items) { r.ok = false; r.p = 0; }
    if ((uint32_t)r.n >= (uint64_t)T.n_items) { r.ok = false; r.n = 0; }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if ((uint32_t)r.mp[m] >= (uint64_t)T.n_meta[m]) { r.ok = false; r.mp[m] = 0; }
      if ((uint32_t)r.mn[m] >= (uint64_t)T.n_meta[m]) { r.ok = false; r.mn[m] = 0; }
    }
  };
  auto gather = [&](MetaRows<VEC, K, MT>& r, const MetaIds<MT>& id) {  // every row of the triple, nothing consumed
    row_load<VEC, G, K, FULL>(r.u, T.user, id.u, D, lig);
    row_load<VEC, G, K, FULL>(r.pi, T.item, id.p, D, lig);
    row_load<VEC, G, K, FULL>(r.ni, T.item, id.n, D, lig);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      row_load<VEC, G, K, FULL>(r.rp[m], T.meta[m], id.mp[m], D, lig);
      row_load<VEC, G, K, FULL>(r.rn[m], T.meta[m], id.mn[m], D, lig);
    }
    r.ul = T.user_lin[id.u]; r.pl = T.item_lin[id.p]; r.nl = T.item_lin[id.n];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      r.lp[m] = NET == TRS_NET_FM ? T.meta_lin[m][id.mp[m]] : 0.f;
      r.ln[m] = NET == TRS_NET_FM ? T.meta_lin[m][id.mn[m]] : 0.f;
    }
    if (OPT != OPT_SGD) {
      row_load<VEC, G, K, FULL>(r.us1, a.o.user_s1, id.u, D, lig);
      r.uls1 = a.o.user_lin_s1[id.u];
      if (OPT == OPT_ADAM) {
        row_load<VEC, G, K, FULL>(r.us2, a.o.user_s2, id.u, D, lig);
        r.uls2 = a.o.user_lin_s2[id.u];
      }
    }
  };
  auto reduce = [&](MetaRows<VEC, K, MT>& r, const MetaIds<MT>& id, int64_t t) {
    if (id.valid && !id.ok && lig == 0 && a.Bt.err_flag_dev) atomicOr(a.Bt.err_flag_dev, 1);
    const bool live = id.valid && id.ok;
    // the two passes, operation by operation as pass_forward (score_kernels.h) does them
    RowReg<VEC, K> Sp, Sn;
    float sqp[N], sqn[N], lin_p, lin_n;
    if (NET == TRS_NET_FM) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        Sp.v[n] = r.u.v[n] + r.pi.v[n];
        Sn.v[n] = r.u.v[n] + r.ni.v[n];
        sqp[n] = r.u.v[n] * r.u.v[n] + r.pi.v[n] * r.pi.v[n];
        sqn[n] = r.u.v[n] * r.u.v[n] + r.ni.v[n] * r.ni.v[n];
      }
      lin_p = r.ul + r.pl;
      lin_n = r.ul + r.nl;
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        Sp.v[n] = r.pi.v[n];
        Sn.v[n] = r.ni.v[n];
      }
      lin_p = lin_n = 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        Sp.v[n] += r.rp[m].v[n];
        Sn.v[n] += r.rn[m].v[n];
        if (NET == TRS_NET_FM) {
          sqp[n] += r.rp[m].v[n] * r.rp[m].v[n];
          sqn[n] += r.rn[m].v[n] * r.rn[m].v[n];
        }
      }
      if (NET == TRS_NET_FM) {
        lin_p += r.lp[m];
        lin_n += r.ln[m];
      }
    }
    float pp = 0.f, pn = 0.f;
    if (NET == TRS_NET_FM) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        pp += Sp.v[n] * Sp.v[n] - sqp[n];
        pn += Sn.v[n] * Sn.v[n] - sqn[n];
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        pp += r.u.v[n] * Sp.v[n];
        pn += r.u.v[n] * Sn.v[n];
      }
    }
    pp = trs_group_sum<G>(pp);
    pn = trs_group_sum<G>(pn);
    const float sp = NET == TRS_NET_FM ? sigmoidf_(lin_p + 0.5f * pp) : (pp + r.ul) + r.pl;
    const float sn = NET == TRS_NET_FM ? sigmoidf_(lin_n + 0.5f * pn) : (pn + r.ul) + r.nl;
    float lval, dneg;
    trs_pair_loss(a.loss, sp, sn, lval, dneg);
    const float act = live ? dneg : 0.f;
    float gp = -act * a.inv_B, gn = act * a.inv_B;
    if (live && lig == 0) loss_acc += lval;
    if (NET == TRS_NET_FM) {
      gp = gp * ((1.0f - sp) * sp);
      gn = gn * ((1.0f - sn) * sn);
    }
    if (id.valid) {
      const int64_t BD = B * (int64_t)D;
      RowReg<VEC, K> g;
      if (NET == TRS_NET_FM) {
        row_store<VEC, G, K>(Sp, a.xstage + t * (int64_t)D, D, lig);
        row_store<VEC, G, K>(Sn, a.xstage + BD + t * (int64_t)D, D, lig);
#pragma unroll
        for (int n = 0; n < N; ++n) g.v[n] = gp * (Sp.v[n] - r.u.v[n]) + gn * (Sn.v[n] - r.u.v[n]);
      } else {
        row_store<VEC, G, K>(r.u, a.xstage + t * (int64_t)D, D, lig);
#pragma unroll
        for (int n = 0; n < N; ++n) g.v[n] = gp * Sp.v[n] + gn * Sn.v[n];
      }
      if (id.dup || !live) {
        row_store<VEC, G, K>(g, a.du + t * (int64_t)D, D, lig);
      } else {
        RowReg<VEC, K> un;
        if (OPT == OPT_SGD) {
#pragma unroll
          for (int n = 0; n < N; ++n) un.v[n] = r.u.v[n] + (-a.lr) * g.v[n];
          row_store<VEC, G, K>(un, T.user + id.u * (int64_t)D, D, lig);
          if (lig == 0) T.user_lin[id.u] = r.ul + (-a.lr) * (gp + gn);
        } else {
#pragma unroll
          for (int n = 0; n < N; ++n) un.v[n] = opt_apply<OPT>(r.u.v[n], g.v[n], r.us1.v[n], r.us2.v[n], a.o);
          row_store<VEC, G, K>(un, T.user + id.u * (int64_t)D, D, lig);
          row_store<VEC, G, K>(r.us1, a.o.user_s1 + id.u * (int64_t)D, D, lig);
          if (OPT == OPT_ADAM) row_store<VEC, G, K>(r.us2, a.o.user_s2 + id.u * (int64_t)D, D, lig);
          if (lig == 0) {
            T.user_lin[id.u] = opt_apply<OPT>(r.ul, gp + gn, r.uls1, r.uls2, a.o);
            a.o.user_lin_s1[id.u] = r.uls1;
            if (OPT == OPT_ADAM) a.o.user_lin_s2[id.u] = r.uls2;
          }
        }
      }
      if (lig == 0) {
        a.gz[t] = gp;
        a.gz[B + t] = gn;
      }
    }
  };

  // the software pipeline of fwd_stage_kernel: ids two triples ahead, rows one ahead, two register sets by parity
  int64_t t = wave * TPW + lane / G;
  const int64_t t_first = wave * TPW;
  const int64_t niter = t_first < B ? (B - t_first + stride - 1) / stride : 0;
  const int64_t niter2 = (niter + 1) & ~(int64_t)1;
  MetaIds<MT> wE = issue(t), wO = issue(t + stride), idE, idO;
  MetaRows<VEC, K, MT> rE, rO;
  idE = wE;
  finalize(idE);
  gather(rE, idE);
  for (int64_t it = 0; it < niter2; it += 2) {
    wE = issue(t + 2 * stride);
    idO = wO;
    finalize(idO);
    gather(rO, idO);
    reduce(rE, idE, t);

    wO = issue(t + 3 * stride);
    idE = wE;
    finalize(idE);
    gather(rE, idE);
    reduce(rO, idO, t + stride);
    t += 2 * stride;
  }
  __shared__ float s_loss[TRS_BLOCK / TRS_WAVE];
  const float wl = trs_wave_sum(loss_acc);
  if (lane == 0) s_loss[threadIdx.x >> 6] = wl;
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < TRS_BLOCK / TRS_WAVE; ++w) L += s_loss[w];
    if (L != 0.f && a.loss_sum) atomicAdd(a.loss_sum, L);
  }
}

template <int NET, int MT>
static int launch_meta_stage_mt(const ScoreArgs& a, hipStream_t s) {
  RowCfg c;
  if (!pick_row_cfg(a.T.D, c)) {
    trs_set_error("unsupported n_factors D=%d", a.T.D);
    return TRS_E_ARG;
  }
  const int tpw = TRS_WAVE / c.g;
  int64_t iters = (a.Bt.B + 512 * 4 * (int64_t)tpw - 1) / (512 * 4 * (int64_t)tpw);  // as launch_fwd_stage
  iters = iters < 2 ? 2 : (iters > 8 ? 8 : iters);
  int64_t grid = ((a.Bt.B + tpw - 1) / tpw + 4 * iters - 1) / (4 * iters);
  grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
  const dim3 gr((unsigned)grid), bl(TRS_BLOCK);
#define TRS_CASE(V, GG, KK)                                                                                   \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                                 \
    if (a.o.kind != OPT_SGD) { /* adaptive rules: whole-row shapes only */                                    \
      if (V * GG * KK != a.T.D) return 1;                                                                     \
      if (a.o.kind == OPT_ADAM)                                                                               \
        hipLaunchKernelGGL((meta_stage_kernel<NET, V, GG, KK, true, MT, OPT_ADAM>), gr, bl, 0, s, a);          \
      else                                                                                                    \
        hipLaunchKernelGGL((meta_stage_kernel<NET, V, GG, KK, true, MT, OPT_ADAGRAD>), gr, bl, 0, s, a);       \
    } else if (V * GG * KK == a.T.D)                                                                          \
      hipLaunchKernelGGL((meta_stage_kernel<NET, V, GG, KK, true, MT>), gr, bl, 0, s, a);                      \
    else                                                                                                      \
      hipLaunchKernelGGL((meta_stage_kernel<NET, V, GG, KK, false, MT>), gr, bl, 0, s, a);                     \
    TRS_CHECK_LAUNCH("meta_stage_kernel");                                                                    \
    return TRS_OK;                                                                                            \
  }
  TRS_CASE(4, 8, 1)
  TRS_CASE(4, 16, 1)
  TRS_CASE(4, 32, 1)
  TRS_CASE(4, 64, 1)
#undef TRS_CASE
  return 1;  // shape not instantiated: the caller falls back to the generic scorer's staging mode
}

// > 0: not handled here (caller uses score_kernel<MODE 2>)
template <int NET>
static int launch_meta_stage(const ScoreArgs& a, hipStream_t s) {
  if (!a.Bt.pos_meta || !a.Bt.neg_meta || a.grad_rows) return 1;
  switch (a.T.M) {
    case 1: return launch_meta_stage_mt<NET, 1>(a, s);
    case 2: return launch_meta_stage_mt<NET, 2>(a, s);
    case 3: return launch_meta_stage_mt<NET, 3>(a, s);
    default: return 1;
  }
}

// Workgroups of fwd_stage_kernel<..., INL 3> that are certainly resident at once: (occupancy - 1) per CU (one below the
// API's answer, which can be one too high — MI355X_MICROARCH.md "Residency and cooperative launch"), at most 2 per CU.
template <typename KernelT>
static int64_t defer_resident_cap(KernelT kernel) {
  int occ = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, TRS_BLOCK, 0) != hipSuccess) return 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  const int want = trs_tuning().k1_wgs_per_cu;
  const int per_cu = occ - 1 < want ? occ - 1 : want;
  return per_cu > 0 ? (int64_t)per_cu * cus : 0;
}

template <int NET>
static int launch_fwd_stage(const FastArgs& a, hipStream_t s, uint32_t* deferred = nullptr) {
  if (deferred) *deferred = 0;  // > 0: INL 3 was launched with that many workgroups (= arrivals on a.sync)
  RowCfg c;
  if (!pick_row_cfg(a.T.D, c)) {
    trs_set_error("unsupported n_factors D=%d", a.T.D);
    return TRS_E_ARG;
  }
  const int tpw = TRS_WAVE / c.g;
  // ~8 pipelined iterations per lane group: few enough workgroups that every wave overlaps its own loads with its
  // own reductions, enough (>= 2 per CU) to fill the chip
  // measured at c2 (B = 65536, D = 64): 2 iterations 35 us, 4: 26, 6-8: 22-25, 16: 29.  Small batches keep >= 2
  // iterations (the pipeline's minimum) but spread over as many workgroups as there is work for.
  const int iters_env = trs_tuning().k1_iters;
  int64_t iters = iters_env > 0 ? iters_env : (a.B + 512 * 4 * (int64_t)tpw - 1) / (512 * 4 * (int64_t)tpw);
  if (iters < 2) iters = 2;
  if (iters > 8 && iters_env <= 0) iters = 8;
  int64_t grid = ((a.B + tpw - 1) / tpw + 4 * iters - 1) / (4 * iters);
  if (grid < 1) grid = 1;
  if (grid > 4096) grid = 4096;
  const int src = !a.from_stream ? 0 : (a.neg_static ? 2 : 1);
  // one-launch flag mode: user rows loaded AND stored nontemporally when the user table is far beyond the Infinity Cache
  // (as in the scoring pass, trs_launch_pair_scores): c4, 1 024-step windows, 37.0-37.4 -> 36.2-36.3 us per step (loads
  // alone: no change).  TRS_K1_NT = 0 | 1 | 3 forces a setting.
  const int ntu_env = trs_tuning().k1_nt;
  const int ntu = ntu_env >= 0 ? ntu_env : ((int64_t)a.T.n_users * a.T.D * 4 > ((int64_t)512 << 20) ? 3 : 0);
#define TRS_LAUNCH(V, GG, KK, FULL)                                                                             \
  {                                                                                                             \
    const dim3 gr((unsigned)grid), bl(TRS_BLOCK);                                                               \
    if (src == 0 && a.udup_pos && a.o.kind == OPT_ADAM)                                                          \
      hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 1, OPT_ADAM>), gr, bl, 0, s, a);              \
    else if (src == 0 && a.udup_pos && a.o.kind == OPT_ADAGRAD)                                                  \
      hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 1, OPT_ADAGRAD>), gr, bl, 0, s, a);           \
    else if (src == 0 && a.udup_pos && a.idup_pos && a.sync && deferred && iters <= DEFER_ITERS) {               \
      static const int64_t cap = defer_resident_cap(fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 3>);                \
      if (grid <= cap) {                                                                                        \
        FastArgs b = a;                                                                                         \
        b.sync_target = a.sync_base + (uint32_t)grid;                                                           \
        if (ntu == 1) hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 3, OPT_SGD, 1>), gr, bl, 0, s, b); \
        else if (ntu == 3) hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 3, OPT_SGD, 3>), gr, bl, 0, s, b); \
        else hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 3>), gr, bl, 0, s, b);                 \
        *deferred = (uint32_t)grid;                                                                             \
      } else                                                                                                    \
        hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 2>), gr, bl, 0, s, a);                      \
    } else if (src == 0 && a.udup_pos && a.idup_pos)                                                             \
      hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 2>), gr, bl, 0, s, a);                        \
    else if (src == 0 && a.udup_pos) hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 1>), gr, bl, 0, s, a); \
    else if (src == 0) hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 0, FULL, 0>), gr, bl, 0, s, a);      \
    else if (src == 1) hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 1, FULL, 0>), gr, bl, 0, s, a);      \
    else hipLaunchKernelGGL((fwd_stage_kernel<NET, V, GG, KK, 2, FULL, 0>), gr, bl, 0, s, a);                    \
  }
#define TRS_CASE(V, GG, KK)                                                                                     \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                                   \
    if (V * GG * KK == a.T.D) TRS_LAUNCH(V, GG, KK, true) else TRS_LAUNCH(V, GG, KK, false)                      \
    TRS_CHECK_LAUNCH("fwd_stage_kernel");                                                                       \
    return TRS_OK;                                                                                              \
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
#undef TRS_LAUNCH
  trs_set_error("internal: no kernel for D=%d", a.T.D);
  return TRS_E_ARG;
}

static int launch_flagged_update(const FastArgs& a, hipStream_t s) {
  const dim3 gr(trs_grid((3 * a.B + FLG_REFS - 1) / FLG_REFS, TRS_BLOCK / TRS_WAVE)), bl(TRS_BLOCK);
  const int kd = (a.T.D + TRS_WAVE - 1) / TRS_WAVE;
  if (kd <= 1) hipLaunchKernelGGL(flagged_update_kernel<1>, gr, bl, 0, s, a);
  else if (kd <= 2) hipLaunchKernelGGL(flagged_update_kernel<2>, gr, bl, 0, s, a);
  else if (kd <= 4) hipLaunchKernelGGL(flagged_update_kernel<4>, gr, bl, 0, s, a);
  else if (kd <= 8) hipLaunchKernelGGL(flagged_update_kernel<8>, gr, bl, 0, s, a);
  else hipLaunchKernelGGL(flagged_update_kernel<16>, gr, bl, 0, s, a);
  TRS_CHECK_LAUNCH("flagged_update_kernel");
  return TRS_OK;
}

template <int WHICH>  // 0: item owners (K2a), 1: users (K3), 2: duplicated users only (K3')
static int launch_plain(const FastArgs& a, hipStream_t s) {
  RowCfg c;
  if (!pick_row_cfg(a.T.D, c)) {
    trs_set_error("unsupported n_factors D=%d", a.T.D);
    return TRS_E_ARG;
  }
  const int per = (TRS_WAVE / c.g) * 2;
  const dim3 gr(trs_grid((a.B + per - 1) / per, TRS_BLOCK / TRS_WAVE)), bl(TRS_BLOCK);
#define TRS_PL(V, GG, KK, FULL)                                                                        \
  {                                                                                                    \
    if (WHICH == 0) hipLaunchKernelGGL((item_owner_update_kernel<V, GG, KK, FULL>), gr, bl, 0, s, a);  \
    else if (WHICH == 1) hipLaunchKernelGGL((user_plain_update_kernel<V, GG, KK, FULL>), gr, bl, 0, s, a); \
    else hipLaunchKernelGGL((user_dup_update_kernel<V, GG, KK, FULL>), gr, bl, 0, s, a);               \
  }
#define TRS_CASE(V, GG, KK)                                                                     \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                   \
    if (V * GG * KK == a.T.D) TRS_PL(V, GG, KK, true) else TRS_PL(V, GG, KK, false)             \
    TRS_CHECK_LAUNCH("plain update kernel");                                                    \
    return TRS_OK;                                                                              \
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
#undef TRS_PL
  trs_set_error("internal: no kernel for D=%d", a.T.D);
  return TRS_E_ARG;
}

static int launch_updates(const FastArgs& a, hipStream_t s, hipEvent_t ev_k3) {
  const int D = a.T.D;
  if (a.uown) {
    int rc = launch_plain<0>(a, s);  // K2a: owners, plain
    if (rc) return rc;
  }
#define TRS_UPD(L, KD, U)                                                                            \
  {                                                                                                  \
    constexpr int PER = (TRS_WAVE / L) * U;                                                          \
    const dim3 gr(trs_grid((a.B + PER - 1) / PER, TRS_BLOCK / TRS_WAVE)), bl(TRS_BLOCK);             \
    if (a.uown) {                                                                                    \
      hipLaunchKernelGGL((item_update_kernel<L, KD, U, 2>), gr, bl, 0, s, a);                        \
      if (ev_k3) (void)hipEventRecord(ev_k3, s);                                                     \
    } else {                                                                                         \
      hipLaunchKernelGGL((item_update_kernel<L, KD, U, 0>), gr, bl, 0, s, a);                        \
      if (ev_k3) (void)hipEventRecord(ev_k3, s);                                                     \
      hipLaunchKernelGGL((user_update_kernel<L, KD, U, false>), gr, bl, 0, s, a);                    \
    }                                                                                                \
  }
  if (D > 256) TRS_UPD(64, 16, 1)
  else if (D > 128) TRS_UPD(64, 4, 2)
  else if (D > 64) TRS_UPD(64, 2, 4)
  else if (D > 32) TRS_UPD(64, 1, 4)
  else if (D > 16) TRS_UPD(32, 1, 2)
  else if (D > 8) TRS_UPD(16, 1, 2)
  else if (D > 4) TRS_UPD(8, 1, 1)
  else if (D > 1) TRS_UPD(4, 1, 1)
  else TRS_UPD(1, 1, 1)
#undef TRS_UPD
  TRS_CHECK_LAUNCH("item/user_update_kernel");
  if (a.uown) return launch_plain<1>(a, s);  // K3: users, plain (atomics for the few duplicated rows)
  return TRS_OK;
}

}  // namespace trs

using namespace trs;

// presort.hip
int trs_launch_sorted_item_update(const trs_tables* tables, const void* keys_step, const void* vals_step, int key_bytes,
                                  int64_t batch, int64_t item_bits, const float* gz, float lr, uint64_t* uown,
                                  uint32_t* udup, uint32_t stamp, const float* ustage, const int32_t* user_ids,
                                  hipStream_t s);
int trs_item_bits_for(int64_t n_items);
int trs_launch_sorted_updates_fused(const trs_tables* tables, const void* keys_step, const void* vals_step,
                                    int64_t batch, int64_t item_bits, const float* gz, float lr, const float* ustage,
                                    const void* ukeys_step, const void* uvals_step, int64_t q0, const float* du,
                                    const OptArgs* opt, int parity, int64_t xpass, int fmsub, int skip_single,
                                    const uint8_t* uflags_step, const int32_t* user_ids_step, hipStream_t s);
int trs_launch_sorted_meta_update(const trs_tables* tables, int m, float* lin_or_scratch, const void* keys_step,
                                  const void* vals_step, int64_t batch, const float* gz, float lr, const float* xstage,
                                  int64_t xpass, int fmsub, const OptArgs* opt, int parity, hipStream_t s);
// rows.hip
int trs_launch_sgd_fields(int net, const trs_tables* tables, const trs_batch* batch, const float* grad_rows_dev,
                          const float* grad_lin_dev, float lr, int f_begin, void* stream);
int trs_launch_sorted_user_dup_update(const trs_tables* tables, const void* ukeys_step, const void* uvals_step,
                                      int key_bytes, int64_t batch, int64_t q0, const float* du, const float* gz,
                                      float lr, hipStream_t s);

// > 0: shape not handled (caller uses score_kernel<MODE 0>)
int trs_launch_pair_scores(int net, const ScoreArgs* ap, hipStream_t s) {
  const ScoreArgs& a = *ap;
  if (a.T.M != 0 || a.Bt.idx_bytes != 4 || !a.Bt.neg || !a.neg_score || a.iota_user >= 0 || !a.T.user_lin ||
      !a.T.item_lin || a.Bt.B < 1)
    return 1;
  RowCfg c;
  if (!pick_row_cfg(a.T.D, c) || c.vec != 4 || c.k != 1 || c.g < 8) return 1;
  const int tpw = TRS_WAVE / c.g;
  const int iters_env = trs_tuning().pass_iters;
  int64_t iters = iters_env > 0 ? iters_env : (a.Bt.B + 512 * 4 * (int64_t)tpw - 1) / (512 * 4 * (int64_t)tpw);
  iters = iters < 2 ? 2 : (iters > 8 && iters_env <= 0 ? 8 : iters);
  int64_t grid = ((a.Bt.B + tpw - 1) / tpw + 4 * iters - 1) / (4 * iters);
  const int64_t grid_cap = trs_tuning().pass_grid_cap;
  grid = grid < 1 ? 1 : (grid > grid_cap ? grid_cap : grid);
  const dim3 gr((unsigned)grid), bl(TRS_BLOCK);
  // user rows nontemporal when the user table is far beyond the Infinity Cache (256 MiB): its rows are read once per
  // pass and would only displace the item rows and 1-wide terms that do come back.  Measured at c4 (10M x 128 fp32 = 5.1
  // GB of user rows, 512 MB of item rows), fresh triples per launch: B = 262 144 75.7 -> 74.2 us (0.680 -> 0.694 of the
  // HBM peak), 64 batches per launch 0.712 -> 0.723-0.729; item rows nontemporal as well: 0.663 (TRS_PASS_NT = 0..3
  // forces a setting: bit 0 user rows, bit 1 item rows)
  const int ntm_env = trs_tuning().pass_nt;
  const int ntm = ntm_env >= 0 ? ntm_env : ((int64_t)a.T.n_users * a.T.D * 4 > ((int64_t)512 << 20) ? 1 : 0);
#define TRS_PS(NETV, V, GG)                                                                                    \
  {                                                                                                            \
    if (V * GG == a.T.D && ntm == 1) hipLaunchKernelGGL((pair_scores_kernel<NETV, V, GG, 1, true, 1>), gr, bl, 0, s, a); \
    else if (V * GG == a.T.D && ntm == 2) hipLaunchKernelGGL((pair_scores_kernel<NETV, V, GG, 1, true, 2>), gr, bl, 0, s, a); \
    else if (V * GG == a.T.D && ntm == 3) hipLaunchKernelGGL((pair_scores_kernel<NETV, V, GG, 1, true, 3>), gr, bl, 0, s, a); \
    else if (V * GG == a.T.D) hipLaunchKernelGGL((pair_scores_kernel<NETV, V, GG, 1, true>), gr, bl, 0, s, a); \
    else hipLaunchKernelGGL((pair_scores_kernel<NETV, V, GG, 1, false>), gr, bl, 0, s, a);                     \
  }
#define TRS_CASE(GG)                                                          \
  if (c.g == GG) {                                                            \
    if (net == TRS_NET_FM) TRS_PS(TRS_NET_FM, 4, GG) else TRS_PS(TRS_NET_LINEAR, 4, GG) \
    TRS_CHECK_LAUNCH("pair_scores_kernel");                                   \
    return TRS_OK;                                                            \
  }
  TRS_CASE(8)
  TRS_CASE(16)
  TRS_CASE(32)
  TRS_CASE(64)
#undef TRS_CASE
#undef TRS_PS
  return 1;
}

extern "C" int64_t trs_train_scratch_bytes(int64_t n_users, int64_t n_items, int64_t batch, int32_t D) {
  if (n_users <= 0 || n_items <= 0 || batch <= 0 || D <= 0) return 0;
  return 12 * (n_users + n_items);  // uown, iown (8 B per row) + udup, idup (4 B per row)
}

extern "C" int trs_train_steps_sgd(const trs_train_args* args, void* stream) {
  TRS_REQUIRE(args != nullptr, "trs_train_steps_sgd: NULL arguments");
  const int net = args->net;
  const trs_tables* tables = args->tables;
  const int32_t* stream_ui_dev = args->stream_ui_dev;
  const int32_t* neg_static_dev = args->neg_static_dev;
  const int64_t N = args->N;
  const uint64_t shuffle_key = args->shuffle_key, sample_seed = args->sample_seed;
  const int64_t first_pos = args->first_pos, batch = args->batch;
  const int32_t n_steps = args->n_steps;
  const float lr = args->lr;
  int32_t *user_buf_dev = args->user_buf_dev, *pos_buf_dev = args->pos_buf_dev, *neg_buf_dev = args->neg_buf_dev;
  float *gz_buf_dev = args->gz_buf_dev, *du_buf_dev = args->du_buf_dev, *loss_sums_dev = args->loss_sums_dev;
  int32_t* err_flag_dev = args->err_flag_dev;
  void* scratch_dev = args->scratch_dev;
  const uint32_t first_stamp = args->first_stamp;
  const void *sorted_keys_dev = args->sorted_keys_dev, *sorted_vals_dev = args->sorted_vals_dev;
  const int32_t key_bytes = args->key_bytes, ukey_bytes = args->ukey_bytes;
  const uint8_t* user_dup_flags_dev = args->user_dup_flags_dev;
  float* ustage_buf_dev = args->ustage_buf_dev;
  const void *sorted_ukeys_dev = args->sorted_ukeys_dev, *sorted_uvals_dev = args->sorted_uvals_dev;
  const int64_t slice_pos0 = args->slice_pos0;
  const trs_opt* opt = args->opt;
  const trs_meta_stage* meta = args->meta;
  void** events = args->events;
  TRS_REQUIRE(net == TRS_NET_LINEAR || net == TRS_NET_FM, "trs_train_steps_sgd: bad net");
  TRS_REQUIRE(tables && tables->M >= 0 && tables->M <= TRS_MAX_META, "trs_train_steps_sgd: bad M");
  TRS_REQUIRE((tables->M > 0) == (meta != nullptr), "trs_train_steps_sgd: metadata tables need the metadata staging");
  TRS_REQUIRE(tables->user && tables->item && tables->user_lin && tables->item_lin, "trs_train_steps_sgd: NULL table");
  TRS_REQUIRE(batch > 0 && batch < ((int64_t)1 << 30) && n_steps >= 0, "trs_train_steps_sgd: bad batch / n_steps");
  TRS_REQUIRE(user_buf_dev && pos_buf_dev && neg_buf_dev && gz_buf_dev && du_buf_dev && loss_sums_dev,
              "trs_train_steps_sgd: NULL buffer");
  TRS_REQUIRE(!scratch_dev || (first_stamp != 0 && (uint64_t)first_stamp + (uint64_t)n_steps < 0xFFFFFFFFull),
              "trs_train_steps_sgd: stamps must be non-zero and must not wrap (zero the scratch and restart at 1)");
  const bool from_stream = stream_ui_dev != nullptr;
  const bool sorted = sorted_keys_dev != nullptr;
  const bool inl = sorted && user_dup_flags_dev != nullptr;
  if (sorted) {
    TRS_REQUIRE(!from_stream && scratch_dev && sorted_vals_dev && (key_bytes == 4 || key_bytes == 8),
                "trs_train_steps_sgd: the presorted mode needs the epoch's id arrays, scratch and sorted references");
    TRS_REQUIRE(!inl || (ustage_buf_dev && ((sorted_ukeys_dev && sorted_uvals_dev && (ukey_bytes == 4 || ukey_bytes == 8)) ||
                                            (!sorted_ukeys_dev && !(opt && opt->kind != TRS_OPT_SGD) && !meta && key_bytes == 4))),
                "trs_train_steps_sgd: user-duplicate flags need the staging buffer and the slice's sorted (batch,user) "
                "pairs (plain SGD without metadata may omit the pairs: flagged users then take float atomics)");
  }
  const bool adaptive = opt && opt->kind != TRS_OPT_SGD;
  // K1 takes the item references that are alone on their row: plain SGD without metadata on the fused two-launch step
  // flag mode (sparse regime): both flag arrays, no sorted references — K1 takes every reference that is alone on its
  // row, the flagged ones follow in flagged_update_kernel
  const bool flgm = !sorted && user_dup_flags_dev && args->item_dup_flags_dev;
  if (flgm)
    TRS_REQUIRE(!from_stream && !adaptive && !meta && ustage_buf_dev,
                "trs_train_steps_sgd: the flag mode takes given ids, both flag arrays and the (batch,D) staging buffer "
                "(plain SGD, no metadata, no sorted references)");
  const bool item_inl = args->item_dup_flags_dev && inl && !adaptive && !meta && key_bytes == 4 &&
                        (ukey_bytes == 4 || !sorted_ukeys_dev);
  TRS_REQUIRE(!args->item_dup_flags_dev || item_inl || flgm,
              "trs_train_steps_sgd: item-duplicate flags need the presorted plain-SGD step without metadata");
  if (meta) {
    TRS_REQUIRE(sorted_keys_dev && user_dup_flags_dev && key_bytes == 4 && ukey_bytes == 4,
                "trs_train_steps_sgd: metadata scorers run on the presorted step");
    const bool meta_sorted = meta->sorted_keys[0] != nullptr;
    if (adaptive) {
      const int D = tables->D;
      TRS_REQUIRE(meta_sorted && meta->pos_meta_ids && meta->neg_meta_ids && tables->M <= 3 &&
                      (D == 32 || D == 64 || D == 128 || D == 256),
                  "trs_train_steps_sgd: adaptive rules on metadata scorers need sorted columns with their id arrays, "
                  "M <= 3 and D in {32, 64, 128, 256}");
      for (int m = 0; m < tables->M; ++m)
        TRS_REQUIRE(opt->meta_s1[m] && opt->meta_lin_s1[m] && opt->meta_gacc[m] && opt->meta_gacc_lin[m] &&
                        opt->meta_cut_rows[m] && opt->meta_cut_count[m] &&
                        (opt->kind != TRS_OPT_SPARSE_ADAM || (opt->meta_s2[m] && opt->meta_lin_s2[m])),
                    "trs_train_steps_sgd: optimiser state of metadata column %d is NULL", m);
    }
    TRS_REQUIRE(meta->item_meta_tab && meta->xstage, "trs_train_steps_sgd: metadata staging buffer is NULL");
    TRS_REQUIRE(meta_sorted || (meta->grad_rows && meta->grad_lin && meta->meta_ids),
                "trs_train_steps_sgd: metadata gradient staging is NULL");
    for (int m = 0; m < tables->M && meta_sorted; ++m)
      TRS_REQUIRE(meta->sorted_keys[m] && meta->sorted_vals[m] && (net == TRS_NET_FM || meta->lin_scratch),
                  "trs_train_steps_sgd: sorted references of metadata column %d are NULL", m);
    for (int m = 0; m < tables->M; ++m)
      TRS_REQUIRE(tables->meta[m] && (net != TRS_NET_FM || tables->meta_lin[m]),
                  "trs_train_steps_sgd: metadata table %d is NULL", m);
  }
  if (adaptive) {
    TRS_REQUIRE(opt->kind == TRS_OPT_SPARSE_ADAM || opt->kind == TRS_OPT_ADAGRAD, "trs_train_steps_sgd: bad opt->kind");
    TRS_REQUIRE(inl && key_bytes == 4 && ukey_bytes == 4,
                "trs_train_steps_sgd: the adaptive rules need the presorted mode with user-duplicate flags");
    TRS_REQUIRE(opt->user_s1 && opt->item_s1 && opt->user_lin_s1 && opt->item_lin_s1 && opt->gacc && opt->gacc_lin &&
                    opt->cut_rows && opt->cut_count && opt->cut_capacity > 0 && opt->step0 >= 0,
                "trs_train_steps_sgd: optimiser state / cut-run scratch is NULL");
    TRS_REQUIRE(opt->kind != TRS_OPT_SPARSE_ADAM ||
                    (opt->user_s2 && opt->item_s2 && opt->user_lin_s2 && opt->item_lin_s2),
                "trs_train_steps_sgd: SparseAdam needs exp_avg_sq tables");
  }
  if (from_stream) {
    TRS_REQUIRE(N > 0 && first_pos >= 0 && first_pos + (int64_t)n_steps * batch <= N,
                "trs_train_steps_sgd: steps [%lld, %lld) outside the stream of %lld rows", (long long)first_pos,
                (long long)(first_pos + (int64_t)n_steps * batch), (long long)N);
    TRS_REQUIRE(neg_static_dev || tables->n_items >= 2, "trs_train_steps_sgd: dynamic sampling needs n_items >= 2");
  }  // without a stream the id buffers hold n_steps consecutive batches (one epoch slice prepared by the host)
  hipStream_t s = (hipStream_t)stream;
  FastArgs a = {};
  a.T = *tables;
  a.user = user_buf_dev;
  a.pos = pos_buf_dev;
  a.neg = neg_buf_dev;
  a.B = batch;
  a.err = err_flag_dev;
  a.from_stream = from_stream ? 1 : 0;
  a.sui = (const int2*)stream_ui_dev;
  a.neg_static = neg_static_dev;
  a.N = N;
  a.shuffle_key = shuffle_key;
  a.hb = from_stream ? trs_feistel_half_bits(N) : 0;
  a.sample_seed = sample_seed;
  a.gz = gz_buf_dev;
  a.du = du_buf_dev;
  a.inv_B = 1.0f / (float)batch;
  a.lr = lr;
  a.loss = args->loss;
  TRS_REQUIRE(args->loss == TRS_LOSS_HINGE || args->loss == TRS_LOSS_BPR, "trs_train_steps_sgd: bad loss kind");
  if (scratch_dev) {
    a.uown = (uint64_t*)scratch_dev;
    a.iown = a.uown + tables->n_users;
    a.udup = (uint32_t*)(a.iown + tables->n_items);
    a.idup = a.udup + tables->n_users;
    if (sorted) a.iown = nullptr;  // K1 marks users only
  }
  const int item_bits = trs_item_bits_for(tables->n_items);
  for (int32_t st = 0; st < n_steps; ++st) {
    a.t0 = first_pos + (int64_t)st * batch;
    a.sample_offset = (uint64_t)a.t0;
    if (inl) {
      a.udup_pos = user_dup_flags_dev + (int64_t)st * batch;
      a.ustage = ustage_buf_dev;
      if (item_inl) a.idup_pos = args->item_dup_flags_dev + (int64_t)st * 2 * batch;
    }
    if (flgm) {
      a.udup_pos = user_dup_flags_dev + (int64_t)st * batch;
      a.idup_pos = args->item_dup_flags_dev + (int64_t)st * 2 * batch;
      a.ustage = ustage_buf_dev;
      a.sync = args->sync_count_host ? args->sync_dev : nullptr;
      a.sync_base = a.sync ? *args->sync_count_host : 0u;
      a.nflag = args->n_flagged_dev ? args->n_flagged_dev + st : nullptr;
    }
    if (adaptive) {  // this step's effective learning rate, in double like the Python floats of torch.optim
      OptArgs& o = a.o;
      const double t = (double)(opt->step0 + st + 1);
      o.kind = opt->kind == TRS_OPT_SPARSE_ADAM ? OPT_ADAM : OPT_ADAGRAD;
      o.beta1 = opt->beta1; o.beta2 = opt->beta2; o.eps = opt->eps;
      o.lr_eff = o.kind == OPT_ADAM
                     ? (float)((double)opt->lr * sqrt(1.0 - pow((double)opt->beta2, t)) / (1.0 - pow((double)opt->beta1, t)))
                     : (float)((double)opt->lr / (1.0 + (t - 1.0) * (double)opt->lr_decay));
      o.user_s1 = opt->user_s1; o.user_s2 = opt->user_s2; o.item_s1 = opt->item_s1; o.item_s2 = opt->item_s2;
      o.user_lin_s1 = opt->user_lin_s1; o.user_lin_s2 = opt->user_lin_s2;
      o.item_lin_s1 = opt->item_lin_s1; o.item_lin_s2 = opt->item_lin_s2;
      o.gacc = opt->gacc; o.gacc_lin = opt->gacc_lin;
      o.cut_rows = opt->cut_rows; o.cut_count = opt->cut_count; o.cut_capacity = opt->cut_capacity;
    }
    if (!from_stream) {  // step st reads its ids at [st*batch, (st+1)*batch) of the given arrays
      a.user = user_buf_dev + (int64_t)st * batch;
      a.pos = pos_buf_dev + (int64_t)st * batch;
      a.neg = neg_buf_dev + (int64_t)st * batch;
    }
    a.loss_sum = loss_sums_dev + st;
    a.stamp = first_stamp + (uint32_t)st;
    hipEvent_t* ev = events ? (hipEvent_t*)events + 4 * (int64_t)st : nullptr;  // K1 | K2a+K2b | K3 boundaries
    if (ev && !ev[0]) ev = nullptr;  // a step whose four handles are NULL is not timed (sampled timing)
    if (ev) (void)hipEventRecord(ev[0], s);
    int rc;
    uint32_t deferred = 0;
    if (meta) {  // K1 = the generic scorer in its staging mode (score_kernels.h MODE 2)
      ScoreArgs sa = {};
      sa.T = *tables;
      sa.Bt.user = a.user; sa.Bt.pos = a.pos; sa.Bt.neg = a.neg;
      sa.Bt.B = batch; sa.Bt.idx_bytes = 4; sa.Bt.err_flag_dev = err_flag_dev;
      if (meta->pos_meta_ids && meta->neg_meta_ids) {
        sa.Bt.pos_meta = (void*)(meta->pos_meta_ids + (int64_t)st * batch * tables->M);
        sa.Bt.neg_meta = (void*)(meta->neg_meta_ids + (int64_t)st * batch * tables->M);
      }
      sa.inv_B = a.inv_B;
      sa.loss = a.loss;
      sa.loss_sum = a.loss_sum;
      const bool meta_sorted = meta->sorted_keys[0] != nullptr;
      sa.grad_rows = meta_sorted ? nullptr : meta->grad_rows;
      sa.grad_lin = meta_sorted ? nullptr : meta->grad_lin;
      sa.iota_user = -1;
      sa.item_meta_tab = meta->item_meta_tab;
      sa.gz = a.gz; sa.du = a.du; sa.xstage = meta->xstage;
      sa.udup_pos = a.udup_pos;
      sa.lr = a.lr;
      sa.meta_ids_out = meta->meta_ids;
      if (adaptive) sa.o = a.o;
      rc = net == TRS_NET_FM ? launch_meta_stage<TRS_NET_FM>(sa, s) : launch_meta_stage<TRS_NET_LINEAR>(sa, s);
      if (rc > 0 && adaptive) {
        trs_set_error("trs_train_steps_sgd: no adaptive metadata kernel for this shape");
        return TRS_E_ARG;
      }
      if (rc > 0)  // more than 3 columns / odd D / no id arrays: the generic scorer's staging mode
        rc = net == TRS_NET_FM ? launch_score<TRS_NET_FM, 2>(sa, s) : launch_score<TRS_NET_LINEAR, 2>(sa, s);
    } else {
      rc = net == TRS_NET_FM ? launch_fwd_stage<TRS_NET_FM>(a, s, &deferred)
                             : launch_fwd_stage<TRS_NET_LINEAR>(a, s, &deferred);
    }
    if (rc) return rc;
    if (ev) (void)hipEventRecord(ev[1], s);
    if (flgm) {  // the flagged references: float atomics into the tables (every read of the step is behind us)
      if (deferred) *args->sync_count_host += deferred;  // K1's own workgroups did it after their grid-wide wait
      else rc = launch_flagged_update(a, s);
      if (rc) return rc;
      if (ev) {
        (void)hipEventRecord(ev[2], s);
        (void)hipEventRecord(ev[3], s);
      }
      continue;
    }
    if (sorted) {  // K2: per-run owner update from the presorted references, then K3
      const char* ks = (const char*)sorted_keys_dev + (int64_t)st * 2 * batch * key_bytes;
      const char* vs = (const char*)sorted_vals_dev + (int64_t)st * 2 * batch * 4;
      if (inl && key_bytes == 4 && (ukey_bytes == 4 || !sorted_ukeys_dev)) {  // item + duplicated-user updates in one launch
        const char* uk = sorted_ukeys_dev ? (const char*)sorted_ukeys_dev + (int64_t)st * batch * 4 : nullptr;
        const char* uv = sorted_ukeys_dev ? (const char*)sorted_uvals_dev + (int64_t)st * batch * 4 : nullptr;
        const bool fm_meta = meta && net == TRS_NET_FM;
        rc = trs_launch_sorted_updates_fused(tables, ks, vs, batch, item_bits, a.gz, a.lr,
                                             meta ? meta->xstage : a.ustage, uk, uv,
                                             slice_pos0 + (int64_t)st * batch, a.du, adaptive ? &a.o : nullptr,
                                             (int)(a.stamp & 1u), fm_meta ? batch : 0, fm_meta ? 1 : 0,
                                             item_inl ? 1 : 0, a.udup_pos, a.user, s);
        if (rc) return rc;
        if (meta && meta->sorted_keys[0]) {  // one sorted-run launch per metadata column
          for (int m = 0; m < tables->M; ++m) {
            const char* mk = (const char*)meta->sorted_keys[m] + (int64_t)st * 2 * batch * 4;
            const char* mv = (const char*)meta->sorted_vals[m] + (int64_t)st * 2 * batch * 4;
            OptArgs mo = a.o;  // this column's state, accumulators and cut-run list in the item slots
            if (adaptive) {
              mo.item_s1 = opt->meta_s1[m]; mo.item_s2 = opt->meta_s2[m];
              mo.item_lin_s1 = opt->meta_lin_s1[m]; mo.item_lin_s2 = opt->meta_lin_s2[m];
              mo.gacc = opt->meta_gacc[m]; mo.gacc_lin = opt->meta_gacc_lin[m];
              mo.cut_rows = opt->meta_cut_rows[m]; mo.cut_count = opt->meta_cut_count[m];
            }
            rc = trs_launch_sorted_meta_update(tables, m, net == TRS_NET_FM ? tables->meta_lin[m] : meta->lin_scratch, mk,
                                               mv, batch, a.gz, a.lr, meta->xstage, fm_meta ? batch : 0, fm_meta ? 1 : 0,
                                               adaptive ? &mo : nullptr, (int)(a.stamp & 1u), s);
            if (rc) return rc;
          }
        } else if (meta) {  // the metadata fields staged by K1: atomic scatter into their (small) tables
          trs_batch mb = {};
          mb.user = a.user; mb.pos = a.pos; mb.neg = a.neg;
          mb.pos_meta = meta->meta_ids;
          mb.neg_meta = meta->meta_ids + batch * tables->M;
          mb.B = batch; mb.idx_bytes = 4; mb.err_flag_dev = err_flag_dev;
          rc = trs_launch_sgd_fields(net, tables, &mb, meta->grad_rows, meta->grad_lin, a.lr, 3, s);
          if (rc) return rc;
        }
        if (ev) {
          (void)hipEventRecord(ev[2], s);
          (void)hipEventRecord(ev[3], s);
        }
        continue;
      }
      rc = trs_launch_sorted_item_update(tables, ks, vs, key_bytes, batch, item_bits, a.gz, a.lr,
                                         inl ? nullptr : a.uown, a.udup, a.stamp, inl ? a.ustage : nullptr, a.user, s);
      if (rc) return rc;
      if (ev) (void)hipEventRecord(ev[2], s);
      if (inl) {
        const char* uk = (const char*)sorted_ukeys_dev + (int64_t)st * batch * ukey_bytes;
        const char* uv = (const char*)sorted_uvals_dev + (int64_t)st * batch * 4;
        rc = trs_launch_sorted_user_dup_update(tables, uk, uv, ukey_bytes, batch, slice_pos0 + (int64_t)st * batch,
                                               a.du, a.gz, a.lr, s);
      } else {
        rc = launch_plain<1>(a, s);
      }
    } else {
      rc = launch_updates(a, s, ev ? ev[2] : nullptr);
    }
    if (rc) return rc;
    if (ev) (void)hipEventRecord(ev[3], s);
  }
  return TRS_OK;
}
