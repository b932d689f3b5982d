/*
 * trs.h — C-ABI of libtrs_hip.so: the MI355X (gfx950) hot path of the torchrecsys training loop.
 *
 * The reference (FrancescoI/torchrecsys) has no FFI; the seams this library sits behind are Python call
 * contracts.  Every entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add to bind them.
 *
 * Conventions
 *  - every pointer named *_dev / inside trs_tables / trs_batch is a DEVICE pointer owned by the caller
 *    (PyTorch-ROCm allocates; the library never allocates, frees or synchronises);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream) and is
 *    safe to capture into a hipGraph;
 *  - return value: 0 = OK, negative = TRS_E_* ; trs_last_error() gives the message of the calling thread's
 *    last failure.  No C++ exception crosses the boundary;
 *  - ids are int32 or int64 (`idx_bytes` = 4 or 8): the reference uses int64 everywhere
 *    (dataset/dataset.py:268-269), the device-resident interaction stream uses int32;
 *  - an id outside its table is never dereferenced: the triple is skipped and bit 0 of *err_flag_dev is set
 *    (the reference raises IndexError from aten::embedding; the host mirror raises it at the next sync);
 *  - "field" order of every per-triple staging array: 0 = user, 1 = pos item, 2 = neg item,
 *    3+2m = pos metadata column m, 4+2m = neg metadata column m  (R = 3 + 2M fields).
 */
#ifndef TRS_H
#define TRS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRS_MAX_META 8
#define TRS_MAX_LAYERS 8

#define TRS_OK 0
#define TRS_E_ARG (-1)     /* bad argument (shape, NULL, unsupported size) */
#define TRS_E_LAUNCH (-2)  /* HIP launch / runtime error */
#define TRS_E_DEVICE (-3)  /* no gfx950 device / wrong architecture */

#define TRS_NET_LINEAR 0
#define TRS_NET_FM 1

/* Pairwise training loss over (positive score, negative score), mean over the batch.  HINGE is the reference's
 * (helper/loss.py:5-9: clamp(neg - pos + 1, min=0)); BPR = -log sigmoid(pos - neg) is the alternative BASELINE.json's
 * north_star names — the reference has no such loss and fit() no loss argument, so BPR is pinned by its formula
 * (oracle/nets.py::bpr_loss against torch autograd), not by reference outputs. */
#define TRS_LOSS_HINGE 0
#define TRS_LOSS_BPR 1

/* Embedding tables of one scorer.  Row-major (n_rows, D) fp32, as nn.Embedding.weight
 * (embeddings/init_embeddings.py:5-50,53-97).
 *   Linear (collaborative/linear.py:43-51): user,item = self.user,self.item; user_lin,item_lin = user_bias,item_bias
 *                                           (n,1); meta[m] = self.metadata[m]; meta_lin unused (NULL).
 *   FM     (collaborative/fm.py:42-56):     user,item; user_lin,item_lin = linear_user,linear_item (n,1);
 *                                           meta[m] = metadata[m]; meta_lin[m] = linear_metadata[m] (n,1).
 *   MLP    (collaborative/mlp.py:66-71):    user,item; meta[m] = metadata_embeddings[m]; *_lin unused. */
typedef struct trs_tables {
  float* user;
  float* item;
  float* user_lin;
  float* item_lin;
  float* meta[TRS_MAX_META];
  float* meta_lin[TRS_MAX_META];
  int64_t n_users;
  int64_t n_items;
  int64_t n_meta[TRS_MAX_META];
  int32_t D; /* n_factors */
  int32_t M; /* metadata columns, 0..TRS_MAX_META */
} trs_tables;

/* One mini-batch as FastDataLoader.__next__ yields it (dataset/dataset.py:414-458):
 * keys user_id, pos_item_id, neg_item_id (B,), pos_metadata_id, neg_metadata_id (B,M) row-major. */
typedef struct trs_batch {
  const void* user;
  const void* pos;
  const void* neg;      /* NULL: score the positive pass only (predict, model.py:436-439) */
  const void* pos_meta; /* (B,M) or NULL when M == 0 */
  const void* neg_meta;
  int64_t B;
  int32_t idx_bytes;     /* 4 or 8 */
  int32_t* err_flag_dev; /* optional; bit 0 set on an out-of-range id */
} trs_batch;

/* ---------------------------------------------------------------------------------------------- misc */
const char* trs_last_error(void);
/* ABI version of this header; bump on ANY signature or struct-layout change.  The ctypes binding refuses a library
 * whose trs_abi_version() differs (torchrecsys_amd/_lib.py::load), tests/test_abi.py checks the three copies agree.
 *   1: round 1.   2: trs_train_steps_sgd takes a trs_train_args struct; trs_epoch_presort writes item-duplicate flags.
 *   3: trs_bn_relu_forward (running statistics, batch counter, output-layer dot) and trs_bn_relu_backward (outer-product
 *      form, outer_xw) grew arguments; trs_hinge_auc_backward, trs_f32_to_bf16_multi, trs_mlp_embed_sgd_update added.
 *   4: trs_sampler.seen_users (bounds of the seen CSR); trs_train_args.sync_dev (flag mode as one launch per step);
 *      trs_mlp_gather_gemm1_fwd, trs_tuning_set added; trs_epoch_flags takes batches up to 262 144; the presort entry
 *      points no longer use their temp buffers (no vendor sort).
 *   5: trs_epoch_flags_ordered (flagged-first batches), trs_train_args.n_flagged_dev. */
#define TRS_ABI_VERSION 5
#define TRS_SYNC_WORDS 288
int trs_abi_version(void);
/* Tuning / A-B knobs of the launch paths (kernel selection, launch shapes): GRID_CAP, PASS_GRID_CAP, K1_ITERS,
 * PASS_ITERS, PRESORT_GRID_CAP, PASS_NT, K1_NT, K1_WGS_PER_CU, GEMM32_NO_GLDS, GEMM16_TN_WIDE, GEMM16_TILE, GEMM16_NO_GLDS, BN_FINAL_TWO_SWEEPS (meanings:
 * csrc/trs_common.h TrsTuning).  The library reads TRS_<name> from the environment ONCE, at its first use; this entry
 * point changes a knob afterwards (tests, tools): unset != 0 restores the default.  The defaults are the measured best. */
int trs_tuning_set(const char* name, int64_t value, int32_t unset);
/* 0 if the current HIP device is gfx950, TRS_E_DEVICE otherwise. */
int trs_check_device(void);

/* HIP timing events owned by the library (bench.py times the dominant kernel with them on the launch stream).
 * create: n new events; elapsed_ms: host-side read after the stream was synchronised. */
int trs_events_create(int32_t n, void** handles_out);
int trs_events_destroy(int32_t n, void** handles);
int trs_events_elapsed_ms(void* start, void* stop, float* ms_out);

/* ------------------------------------------------------------------------- loader / sampler (a9, a10) */
/* Sampler options beyond the reference's (the reference only rejects the row's own positive,
 * dataset/dataset.py:435-447).  NULL everywhere below = the reference's sampler.  Counter-based like the plain sampler
 * (try 0 with every option off IS the plain sampler's draw):
 *   k_neg      every interaction is visited k_neg times per epoch, each time with a fresh negative: the epoch has
 *              N * k_neg positions, position q maps to row perm_{N*k_neg}(q) % N;
 *   popularity candidates are drawn as the item of a uniformly random interaction (pop_items[r], r < pop_n), i.e.
 *              proportionally to the item's frequency in the stream, instead of uniformly over the catalogue;
 *   seen_off / seen_items   CSR of every user's positives (seen_off (n_users+1), seen_items sorted per user): a
 *              candidate the user has interacted with is rejected and redrawn (key = seed + try * golden ratio), at
 *              most max_tries candidates; the last one is kept whatever it is (bounded work).
 * The negative always differs from the row's positive. */
typedef struct trs_sampler {
  int32_t k_neg;      /* >= 1 */
  int32_t popularity; /* 0 | 1 */
  int32_t max_tries;  /* >= 1 */
  int32_t reserved;
  const int64_t* seen_off;
  const int32_t* seen_items;
  const int32_t* pop_items;
  int64_t pop_n;
  int64_t seen_users; /* rows of the seen CSR = n_users; a user id outside [0, seen_users) has seen nothing */
} trs_sampler;

/* Counter-based dynamic negative sampler: neg[t] uniform over {0..n_items-1} \ {pos[t]} — the distribution of the
 * rejection loop at dataset/dataset.py:440-445 (reject only the row's own positive).  Stream: Philox4x32-10,
 * key = seed, counter = offset + t; r = mulhi64(x, n_items-1); neg = r + (r >= pos).  Not the numpy legacy
 * stream (that one is replayed on the host by FastDataLoader for bit-exact reference batches). */
int trs_sample_neg(const void* pos_dev, int idx_bytes, int64_t B, int64_t n_items, uint64_t seed,
                   uint64_t offset, void* neg_out_dev, void* stream);

/* Epoch shuffle + batch slice + negative sampling in one pass over a device-resident interaction stream
 * (replaces torch.randperm + tensor[perm[i:i+B]] at dataset/dataset.py:364-373,420-422 and the sampler loop
 * :435-447).  Row p of the stream is (stream_user[p], stream_item[p]), int32.  Position q = t0 + t of the epoch maps
 * to row perm(q) where perm is a keyed bijection of [0,N) (4-round Feistel network over ceil(log2 N) bits with
 * cycle walking, round function Philox-mix keyed by shuffle_key; shuffle_key = 0 means identity = shuffle=False).
 * Writes user/pos/neg int32 (B,) and, when M > 0, pos_meta/neg_meta (B,M) looked up through item_meta (n_items,M)
 * (the device form of item_to_metadata_map, dataset/dataset.py:375-411).  neg_static (N,) int32 non-NULL selects the
 * static negatives of dataset/dataset.py:56-64 instead of sampling. */
int trs_batch_prepare(const int32_t* stream_user_dev, const int32_t* stream_item_dev,
                      const int32_t* neg_static_dev, int64_t N, uint64_t shuffle_key, int64_t t0, int64_t B,
                      int64_t n_items, uint64_t sample_seed, uint64_t sample_offset,
                      const int32_t* item_meta_dev, int32_t M, int32_t* user_out, int32_t* pos_out,
                      int32_t* neg_out, int32_t* pos_meta_out, int32_t* neg_meta_out, const trs_sampler* sampler,
                      void* stream);

/* ------------------------------------------------------------------ scorers: forward only (a2, a3, a6) */
/* Fused positive+negative scoring pass; the user row is gathered once for both passes.
 * net = TRS_NET_LINEAR: Linear.forward (collaborative/linear.py:54-80), score (B,1);
 * net = TRS_NET_FM:     FM.forward     (collaborative/fm.py:60-101),   score (B,) after sigmoid.
 * Replaces the two net.forward calls of TorchRecSys.forward (model.py:171-185).  neg_score may be NULL iff
 * batch->neg is NULL. */
int trs_score_forward(int net, const trs_tables* tables, const trs_batch* batch, float* pos_score_dev,
                      float* neg_score_dev, void* stream);

/* ------------------------------------------------- scorers: forward + hinge + backward (a2,a3,a5,a6,a7) */
/* One training pass over a batch with all rows read from the PRE-update tables (the semantics of
 * loss.backward() before optimizer.step(), model.py:188-200):
 *   scores as trs_score_forward; h = neg - pos + 1; loss_sum += sum max(h,0) (helper/loss.py:5-9; the caller divides
 *   by B); auc_count += #(pos > neg) (evaluate/metrics.py:23-31); per-triple gradient rows of d(mean hinge)/d(row)
 *   staged field-major into grad_rows (R,B,D) and the 1-wide terms into grad_lin (R,B) — i.e. exactly the
 *   uncoalesced values of the sparse COO gradients autograd builds (EmbeddingBackward), zero rows kept.
 * inv_B = 1/B of the mean.  pos_score/neg_score/auc_count may be NULL.  loss_sum/auc_count are ACCUMULATED
 * (atomic float / int adds): zero them before the first call. */
int trs_score_fwd_bwd(int net, const trs_tables* tables, const trs_batch* batch, float inv_B,
                      float* pos_score_dev, float* neg_score_dev, float* loss_sum_dev,
                      int32_t* auc_count_dev, float* grad_rows_dev, float* grad_lin_dev, int32_t loss, void* stream);

/* Backward only, from upstream d(loss)/d(score) (B,) per pass — the autograd.Function backward used when a caller
 * drives net.forward + its own loss (any objective), same staging as above. */
int trs_score_backward(int net, const trs_tables* tables, const trs_batch* batch, const float* gpos_dev,
                       const float* gneg_dev, float* grad_rows_dev, float* grad_lin_dev, void* stream);

/* ---------------------------------------------------------------- fused SGD training steps (a2,a3,a5-a9) */
/* n_steps whole training steps of a Linear / FM scorer WITHOUT metadata under plain SGD, driven from C (four kernel
 * launches per step, no host work in between) — the inner loop of TorchRecSys.fit (model.py:274-285) for
 * torch.optim.SGD(momentum=0, weight_decay=0).  Step s covers epoch positions [first_pos + s*batch, +batch).
 *   K1  forward + hinge + backward-to-scores at the PRE-update tables (software-pipelined row gathers); derives the
 *       batch from the resident stream (same shuffle / sampler as trs_batch_prepare, sample_offset = epoch position)
 *       when stream_ui (the stream as interleaved int32 pairs {user, item}, (N,2)) is non-NULL and writes it to user/pos/neg_buf (int32, (batch,)); otherwise reads the ids of
 *       step s from user/pos/neg_buf[s*batch ...] (an epoch slice prepared by the host: the bit-exact reference streams).  Stages gz (2,batch) and the user-row gradient du (batch,D);
 *       loss_sums[s] += sum of hinge terms.
 *   K2  item[pos] -= lr*gz+ * user[u], item[neg] -= lr*gz- * user[u] (+ 1-wide item terms) from the unmodified user rows:
 *       K2a the one reference per distinct row whose K1 ownership mark survived, by plain read-modify-write; K2b the
 *       remaining references of duplicated rows, by float atomics.  (One all-atomic launch when scratch is NULL.)
 *   K3  user[u] -= lr*du (+ 1-wide user term): plain for rows referenced once in the step, atomics for the others.
 * Same results as trs_score_fwd_bwd + trs_score_sgd_update (to summation order of duplicate rows).
 * scratch: NULL or trs_train_scratch_bytes(n_users, n_items, batch, D) bytes, zero-initialised once; first_stamp: step counter
 * of the first step, non-zero, strictly increasing over the life of the scratch (re-zero the scratch before it wraps).
 * events: NULL, or 4*n_steps hipEvent_t handles recorded at the K1 | K2a+K2b | K3 boundaries of each step (a step whose handles are NULL is not timed) (bench.py). */
/* Update rule of the presorted two-launch step (opt == NULL: plain SGD with the lr argument).  The adaptive rules are
 * the ones the reference's users can actually run on its sparse embedding gradients:
 * torch.optim.SparseAdam (lazy Adam: rows present in the batch only) and torch.optim.Adagrad's sparse branch, applied
 * to the COALESCED gradient of each row — which the presorted runs provide: users referenced once are updated by K1,
 * duplicated users and item rows by their sorted runs; item runs cut at a 64-reference chunk boundary sum their pieces
 * in gacc and are applied by a small third launch.  State tables have the shapes of the weight tables and are updated
 * in place (the host mirror keeps them in optimizer.state[p], so optimizer.state_dict() stays truthful). */
#define TRS_OPT_SGD 0
#define TRS_OPT_SPARSE_ADAM 1
#define TRS_OPT_ADAGRAD 2
typedef struct trs_opt {
  int32_t kind;
  float lr, beta1, beta2, eps, lr_decay;
  int64_t step0;       /* optimiser step count before the first step of the call (same for the four tables) */
  float *user_s1, *user_s2, *item_s1, *item_s2;                 /* exp_avg | sum, exp_avg_sq | NULL        (n, D) */
  float *user_lin_s1, *user_lin_s2, *item_lin_s1, *item_lin_s2; /* the same for the 1-wide tables          (n, 1) */
  float* gacc;         /* (n_items, D) all-zero between steps: meeting point of the pieces of a cut item run */
  float* gacc_lin;     /* (n_items) */
  int32_t* cut_rows;   /* (cut_capacity) rows with a cut run in the current step */
  int32_t* cut_count;  /* [2] zero-initialised once; the steps alternate between the two counters */
  int32_t cut_capacity; /* >= 2*batch/64 + the number of rows with more than 64 references (2*batch is always enough) */
  /* Metadata scorers (tables->M > 0, trs_meta_stage with sorted columns): the same state per metadata column m < M.
   * meta_lin_* / meta_gacc_lin belong to the 1-wide metadata tables of FM; a Linear scorer has none and passes scratch
   * arrays of n_meta[m] floats.  meta_cut_rows[m]: cut_capacity entries; meta_cut_count[m]: [2], zeroed once. */
  float *meta_s1[TRS_MAX_META], *meta_s2[TRS_MAX_META], *meta_lin_s1[TRS_MAX_META], *meta_lin_s2[TRS_MAX_META];
  float *meta_gacc[TRS_MAX_META], *meta_gacc_lin[TRS_MAX_META];
  int32_t *meta_cut_rows[TRS_MAX_META], *meta_cut_count[TRS_MAX_META];
} trs_opt;
/* Metadata scorers (tables->M > 0) on the presorted step (plain SGD; the adaptive rules of trs_opt with sorted columns,
 * M <= 3 and D in {32, 64, 128, 256}): K1 is the generic scorer in a staging mode that
 * looks the metadata ids up in item_meta_tab, applies the user update in place, stages the rows the sorted item update
 * multiplies (FM: the per-pass field sums, Linear: the user row) and the metadata fields' gradients; user and item rows
 * then go through the sorted runs as without metadata, the (small, heavily shared) metadata tables through an atomic
 * scatter of the staged fields. */
typedef struct trs_meta_stage {
  const int32_t* item_meta_tab; /* (n_items, M) metadata ids of every item */
  float* xstage;                /* FM: (2, batch, D); Linear: (batch, D) */
  float* grad_rows;             /* (3 + 2M, batch, D): fields 3.. are written */
  float* grad_lin;              /* (3 + 2M, batch) */
  int32_t* meta_ids;            /* (2, batch, M) */
  /* Optional: every column's references sorted per batch (trs_epoch_presort_meta, offset to the call's first batch).
   * Then the metadata tables are updated by sorted runs like the item table (no gradient staging, no atomics but for
   * cut runs) and grad_rows / grad_lin / meta_ids may be NULL.  lin_scratch: max(n_meta) floats, used in place of the
   * 1-wide metadata tables a Linear scorer does not have. */
  const void* sorted_keys[TRS_MAX_META];
  const void* sorted_vals[TRS_MAX_META];
  float* lin_scratch;
  /* Optional: the slice's metadata ids by position, (n_steps * batch, M) int32 each, offset to the call's first batch
   * (written by trs_epoch_presort_meta): K1 reads them instead of looking them up behind the item ids. */
  const int32_t* pos_meta_ids;
  const int32_t* neg_meta_ids;
} trs_meta_stage;
/* Sorted references of metadata column m of an epoch slice whose ids exist (trs_epoch_presort): buffers and sizes as
 * trs_epoch_presort_sizes(n_batches, batch, n_cat, ...).  pos/neg_meta_out_dev (both or neither; (n_pos, M) int32): the
 * looked-up ids of column m, written for K1 (trs_meta_stage.pos_meta_ids / neg_meta_ids). */
int trs_epoch_presort_meta(const int32_t* pos_dev, const int32_t* neg_dev, int64_t n_batches, int64_t batch,
                           const int32_t* item_meta_dev, int32_t M, int32_t m, int64_t n_cat, void* keys_dev,
                           void* vals_dev, void* temp_dev, int64_t temp_bytes, int32_t* err_flag_dev,
                           void** sorted_keys_out, void** sorted_vals_out, int32_t* pos_meta_out_dev,
                           int32_t* neg_meta_out_dev, void* stream);
int64_t trs_train_scratch_bytes(int64_t n_users, int64_t n_items, int64_t batch, int32_t D);
/* Arguments of trs_train_steps_sgd (one struct instead of a 33-long positional list: a caller built against another
 * revision fails the trs_abi_version() check instead of shifting device pointers by one slot).  Zero-initialise, then
 * fill the groups that apply. */
typedef struct trs_train_args {
  int32_t net;               /* TRS_NET_LINEAR | TRS_NET_FM */
  int32_t n_steps;
  const trs_tables* tables;
  int64_t batch;
  float lr;                  /* plain SGD (opt == NULL) */
  int32_t loss;              /* TRS_LOSS_HINGE (0, the reference) | TRS_LOSS_BPR */
  uint32_t first_stamp;      /* with scratch: step counter of the first step, non-zero, strictly increasing */
  /* ids: derived from the resident stream (stream_ui != NULL; {user,item} int32 pairs, (N,2)) or given in the buffers */
  const int32_t* stream_ui_dev;
  const int32_t* neg_static_dev;
  int64_t N;
  uint64_t shuffle_key;
  uint64_t sample_seed;
  int64_t first_pos;
  int32_t* user_buf_dev;     /* (batch) outputs when derived; (n_steps*batch) inputs otherwise */
  int32_t* pos_buf_dev;
  int32_t* neg_buf_dev;
  /* staging and results */
  float* gz_buf_dev;         /* (2, batch) */
  float* du_buf_dev;         /* (batch, D) */
  float* loss_sums_dev;      /* (n_steps), accumulated */
  int32_t* err_flag_dev;
  void* scratch_dev;         /* NULL or trs_train_scratch_bytes(...) bytes, zeroed once */
  /* presorted two-launch step (trs_epoch_presort / trs_epoch_user_dups outputs, offset to the call's first batch) */
  const void* sorted_keys_dev;
  const void* sorted_vals_dev;
  int32_t key_bytes;
  int32_t ukey_bytes;
  const uint8_t* user_dup_flags_dev; /* (n_steps*batch) 1: the triple's user has another reference in its batch */
  const uint8_t* item_dup_flags_dev; /* (n_steps*batch, 2) NULL, or per triple {pos, neg}: 1 = the item row has another
                                        reference in the batch.  Given (plain SGD, no metadata): K1 also applies the
                                        item update of every reference that is ALONE on its row in place, and the
                                        sorted-run launch only walks rows referenced more than once.
                                        FLAG MODE (the sparse regime: most rows of a batch referenced once) = both flag
                                        arrays (trs_epoch_flags; they may be conservative), given ids, ustage, and NO
                                        sorted references: the flagged references follow K1 in a small second launch
                                        that adds their contributions into the tables with float atomics (all reads of
                                        the step happened in K1, so this is exact up to the order of those sums) */
  float* ustage_buf_dev;     /* (batch, D) */
  const void* sorted_ukeys_dev;
  const void* sorted_uvals_dev;
  int64_t slice_pos0;
  const trs_opt* opt;        /* NULL: plain SGD */
  const trs_meta_stage* meta;/* NULL iff tables->M == 0 */
  void** events;             /* NULL, or 4*n_steps hipEvent_t handles (a step whose handles are NULL is not timed) */
  uint32_t* sync_dev;        /* NULL, or TRS_SYNC_WORDS (288) uint32 in device memory (an arrival counter and eight flag
                                lines, 128 B apart), zeroed once by the caller, and */
  uint32_t* sync_count_host; /* ... ONE uint32 in HOST memory, zero at that time and owned by this entry point from
                                then on (the arrivals it has scheduled on sync_dev, mod 2^32).  FLAG MODE with both
                                given: whenever every workgroup of K1's grid is resident at once, the step is ONE
                                launch — K1's workgroups count themselves in on sync_dev[0] once their row reads are
                                done, wait for the whole grid, and add the flagged references' contributions
                                themselves (no second launch).  err bit 2: the grid did not become resident within
                                0.2 s (that step's results are not exact). */
  const int32_t* n_flagged_dev; /* NULL, or (n_steps) int32 from trs_epoch_flags_ordered (the id and flag arrays above
                                then are the ones it ordered): batch st's first n_flagged_dev[st] triples are the ones
                                that carry a flagged reference.  The one-launch step then counts a workgroup in as soon
                                as it is past those triples and never waits for the slowest workgroup.  err bit 3: a
                                flagged reference was met behind those triples (arrays of another origin). */
} trs_train_args;
int trs_train_steps_sgd(const trs_train_args* args, void* stream);

/* Epoch-level grouping of the item references by row.  trs_epoch_presort covers n_batches whole batches starting at
 * epoch position first_pos: it writes the triples' ids (generated from the resident stream exactly as
 * trs_batch_prepare would, or taken as given when stream_ui is NULL) and the 2*batch references of every batch sorted
 * by item row (hand-written counting sort in LDS, one workgroup per batch; key = item row (uint32), payload = (t<<1)|which
 * (uint32), t = position inside the batch).  Passing the sorted arrays (offset to the first batch of the call) to
 * trs_train_steps_sgd together with the id arrays replaces K2a/K2b by one atomic-free launch: each run of equal keys is
 * summed by one lane group and applied with a plain whole-row read-modify-write (runs are cut every 64 references; cut
 * pieces of hot rows use float atomics).  Buffers: keys/vals two halves each (sizes from trs_epoch_presort_sizes).
 * item_dup_flags_out_dev (optional, 2*n_batches*batch bytes): byte 2q+w = 1 iff the item row of reference w (0 positive,
 * 1 negative) of position q is referenced again inside q's batch — neighbour compare on the sorted keys, scattered back
 * by payload (trs_train_args.item_dup_flags_dev). */
/* Per-position flags of an epoch slice: 1 iff the triple's user is referenced by another triple of the same batch
 * (per-batch counting sort of the user ids).  Passing them (offset to the first batch) with a (batch,D) staging buffer to
 * trs_train_steps_sgd in presorted mode lets K1 apply the user update itself for users referenced once in the batch
 * (plain store; K1 then stages the OLD user row for the item update instead of the gradient) — K3 shrinks to the
 * duplicated users: with the slice's sorted (user, position) pairs (offset to the call's first batch; slice_pos0 = that
 * batch's first position in the slice) each run of equal users is summed by one lane group and applied with one plain
 * read-modify-write.  The step then contains no float atomic except for cut runs of hot item rows. */
/* The sparse regime's whole presort in ONE launch, no sort: the ids of n_batches whole batches (generated as
 * trs_epoch_presort does, or taken as given) and per-reference duplicate flags — user_dup_flags (n_pos), item_dup_flags
 * (n_pos, 2) — found with a 2^20-bit bitmap in LDS, one 1024-thread workgroup per batch (set / re-set by the later
 * arrivals / test).  Tables with more rows than bits are hashed: the flags are then CONSERVATIVE (never 0 for a shared
 * row, possibly 1 for a lone one), which the flag mode of trs_train_steps_sgd tolerates.  Replaces the shuffle +
 * slicing + sampler loop of dataset/dataset.py:364-373,414-447 for the slice. */
int trs_epoch_flags(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N, uint64_t shuffle_key,
                    uint64_t sample_seed, int64_t first_pos, int64_t n_batches, int64_t batch, int64_t n_users,
                    int64_t n_items, int32_t* user_dev, int32_t* pos_dev, int32_t* neg_dev,
                    uint8_t* user_dup_flags_out_dev, uint8_t* item_dup_flags_out_dev, int32_t* err_flag_dev,
                    const trs_sampler* sampler, void* stream);
/* trs_epoch_flags, and — with n_flagged_out_dev (n_batches int32) — every batch partitioned IN PLACE (ids and flags
 * together; given ids are permuted too) so that the triples with at least one flagged reference come first; their
 * number goes to n_flagged_out_dev[b].  The order of the triples inside a batch means nothing to a training step (its
 * loss and gradients are sums over the batch): the same multiset of triples, the same flags per triple.  A batch with
 * more than 12 288 misplaced triples is left in its order and reported as n_flagged = batch.  Replaces nothing in the
 * reference (its order inside a batch is a random permutation too: FastDataLoader(shuffle=True), model.py:223-225). */
int trs_epoch_flags_ordered(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N, uint64_t shuffle_key,
                            uint64_t sample_seed, int64_t first_pos, int64_t n_batches, int64_t batch, int64_t n_users,
                            int64_t n_items, int32_t* user_dev, int32_t* pos_dev, int32_t* neg_dev,
                            uint8_t* user_dup_flags_out_dev, uint8_t* item_dup_flags_out_dev,
                            int32_t* n_flagged_out_dev, int32_t* err_flag_dev, const trs_sampler* sampler, void* stream);
/* User-duplicate flags alone, by the same LDS bitmap (no sort; conservative for n_users > 2^20).  Plain SGD without
 * metadata needs nothing else about the users: pass the flags to trs_train_steps_sgd with sorted_ukeys_dev = NULL and the
 * flagged users add their staged gradient rows with float atomics in the sorted-run launch. */
int trs_epoch_user_flags(const int32_t* user_dev, int64_t n_batches, int64_t batch, int64_t n_users,
                         uint8_t* flags_out_dev, void* stream);
int trs_epoch_user_dups_sizes(int64_t n_batches, int64_t batch, int64_t n_users, int64_t* ukeys_bytes_out,
                              int64_t* uvals_bytes_out, int64_t* temp_bytes_out);
int trs_epoch_user_dups(const int32_t* user_dev, int64_t n_batches, int64_t batch, int64_t n_users, void* ukeys_dev,
                        void* uvals_dev, void* temp_dev, int64_t temp_bytes, uint8_t* flags_out_dev,
                        void** sorted_ukeys_out, void** sorted_uvals_out, int32_t* ukey_bytes_out, void* stream);
int trs_epoch_presort_sizes(int64_t n_batches, int64_t batch, int64_t n_items, int64_t* key_bytes_out,
                            int64_t* keys_total_bytes_out, int64_t* vals_total_bytes_out, int64_t* temp_bytes_out);
int trs_epoch_presort(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N, uint64_t shuffle_key,
                      uint64_t sample_seed, int64_t first_pos, int64_t n_batches, int64_t batch, int64_t n_users,
                      int64_t n_items, int32_t* user_dev, int32_t* pos_dev, int32_t* neg_dev, void* keys_dev,
                      void* vals_dev, void* temp_dev, int64_t temp_bytes, int32_t* err_flag_dev,
                      void** sorted_keys_out, void** sorted_vals_out, uint8_t* item_dup_flags_out_dev,
                      const trs_sampler* sampler, void* stream);

/* ---------------------------------------------------------------- sparse row optimisers (a7, App. A.5) */
/* table[idx[t]] += alpha * vals[t]  for t < n, rows of D floats, vals row t at vals + t*ld.  Float atomics, one
 * 256-B segment per wave-instruction.  alpha = -lr is torch.optim.SGD's sparse param.add_(grad, alpha=-lr)
 * (third-party torch/optim/sgd.py, called from model.py:198).  Duplicate ids accumulate (uncoalesced COO). */
int trs_rows_scatter_add(float* table_dev, int64_t n_rows, int32_t D, const void* idx_dev, int32_t idx_bytes,
                         const float* vals_dev, int64_t ld, int64_t n, float alpha, int32_t* err_flag_dev,
                         void* stream);

/* Fused SGD update of every table of a scorer from the staging arrays of trs_score_fwd_bwd:
 * table_f[idx_f[t]] -= lr * grad_rows[f][t]  and the 1-wide tables likewise from grad_lin. */
int trs_score_sgd_update(int net, const trs_tables* tables, const trs_batch* batch, const float* grad_rows_dev,
                         const float* grad_lin_dev, float lr, void* stream);

/* Coalescing optimisers (SparseAdam / Adagrad / SGD-momentum need the per-row SUM of duplicates):
 *   1. trs_rows_scatter_add into `acc` (same shape as the table, all-zero between steps);
 *   2. trs_rows_apply_*: for every id in idx, the first arrival per distinct row (elected with atomicExch on
 *      stamp[row] == step_id) applies the update from acc[row] and clears acc[row].  Rows present in the batch are
 *      "touched" even when their gradient is zero.
 * step_id must be unique per (table, step) and never 0 after the stamps were zero-initialised. */
int trs_rows_apply_sparse_adam(float* table_dev, float* acc_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                               int32_t* stamp_dev, int64_t n_rows, int32_t D, const void* idx_dev,
                               int32_t idx_bytes, int64_t n, int32_t step_id, float lr, float beta1, float beta2,
                               float eps, int64_t step_count, void* stream);
int trs_rows_apply_adagrad(float* table_dev, float* acc_dev, float* state_sum_dev, int32_t* stamp_dev,
                           int64_t n_rows, int32_t D, const void* idx_dev, int32_t idx_bytes, int64_t n,
                           int32_t step_id, float clr, float eps, void* stream);

/* ------------------------------------------------------------------------------ hinge / AUC (a5, a12) */
/* loss_sum += sum_t max(neg-pos+1, 0); auc_count += #(pos > neg).  (helper/loss.py:5-9, evaluate/metrics.py:23-31)
 * loss = TRS_LOSS_BPR: the sum of softplus(neg - pos) instead (same for the two functions below). */
int trs_hinge_auc(const float* pos_dev, const float* neg_dev, int64_t B, float* loss_sum_dev,
                  int32_t* auc_count_dev, int32_t loss, void* stream);
/* The same for consecutive batches of `batch` rows of (n_total,) score arrays in one launch: loss_sums_dev[b] /
 * auc_counts_dev[b] (b < ceil(n_total / batch) <= 65535) accumulate batch b's sums — evaluate()'s per-batch metrics
 * (model.py:300-330) without one launch and one host round trip per batch. */
int trs_hinge_auc_batches(const float* pos_dev, const float* neg_dev, int64_t n_total, int64_t batch,
                          float* loss_sums_dev, int32_t* auc_counts_dev, int32_t loss, void* stream);
/* d(mean hinge)/d(pos), d(.)/d(neg): -a/B, +a/B with a = [neg-pos+1 >= 0] (torch clamp subgradient). */
int trs_hinge_backward(const float* pos_dev, const float* neg_dev, int64_t B, float inv_B, float* gpos_dev,
                       float* gneg_dev, int32_t loss, void* stream);
/* trs_hinge_auc and trs_hinge_backward in one sweep over the scores (the MLP training step; the sums are formed in
 * trs_hinge_auc's order, the gradients are trs_hinge_backward's).  loss_sum_dev / auc_count_dev may be NULL. */
int trs_hinge_auc_backward(const float* pos_dev, const float* neg_dev, int64_t B, float inv_B, float* loss_sum_dev,
                           int32_t* auc_count_dev, float* gpos_dev, float* gneg_dev, int32_t loss, void* stream);

/* ---------------------------------------------------------------------------------- predict (a13) */
/* Scores of ONE user against items [item0, item0+n) (model.py:341-452: net.forward over item chunks, pos keys
 * only).  item_meta (n_items,M) int32 gives each item's metadata ids (NULL when M == 0). */
int trs_score_all_items(int net, const trs_tables* tables, int64_t user_id, int64_t item0, int64_t n,
                        const int32_t* item_meta_dev, float* score_out_dev, void* stream);
/* Top-k of scores (n,) by (score descending, index ascending) -> idx_out (k,) int64, any 1 <= k <= n.  workspace: at
 * least trs_topk_workspace_bytes(n, k) bytes.  (torch.sort(descending=True)[:top_k], model.py:447-450)
 * k <= 2048: per-chunk bitonic selection (4096 keys in LDS, iterated); larger k: a full bitonic sort of the padded keys
 * in global memory (the reference sorts all scores for any top_k). */
int64_t trs_topk_workspace_bytes(int64_t n, int32_t k);
int trs_topk(const float* scores_dev, int64_t n, int32_t k, int64_t* idx_out_dev, void* workspace_dev,
             int64_t workspace_bytes, void* stream);


/* ------------------------------------------------------------------------------------- MLP (a4, a7) */
/* Activations of the MLP are kept for both scoring passes stacked: rows [0,B) = positive pass, rows [B,2B) =
 * negative pass ("passes" = 2).  BatchNorm1d statistics are per pass (the reference calls net.forward twice,
 * model.py:171-185).  Workspaces are caller-allocated fp32 arrays of the size the *_workspace_* helpers return. */

/* x0 = [user[u] | item[i] | meta_0[..] | ...] (collaborative/mlp.py:93-105): positive pass into rows [0,B) of x and,
 * when passes == 2, the negative pass into rows [B,2B); row stride ld >= (2+M)*D elements.  x_dev (fp32) and/or
 * x_bf16_dev (bf16, RNE: the operand image of the bf16-resident GEMMs) — either may be NULL. */
int trs_mlp_gather_concat(const trs_tables* tables, const trs_batch* batch, int32_t passes, float* x_dev,
                          void* x_bf16_dev, int64_t ld, void* stream);

/* MLP layer 0 as ONE launch — the "concat-GEMM": y = [user[u] | item[i] | meta_m[..]] W^T + bias with the embedding gather
 * inside the GEMM's A-operand load (replaces the gathers and the two torch.cat of collaborative/mlp.py:93-105 AND
 * fcs[0] of :107; trs_mlp_gather_concat + trs_gemm_* remain for the shapes this entry point does not take).  Rows as in
 * trs_mlp_gather_concat (passes == 2: positive pass then negative pass).  bf16 = 0: W (N, (2+M)D) fp32, y fp32 — the
 * fp32-MFMA LDS-DMA kernel whose A pieces are addressed by id; bf16 = 1: W = the bf16 image of the weight, y fp32
 * (y_dev) or bf16 (y_bf16_dev), fp32 accumulation — table rows rounded to bf16 (RNE) while they are staged.  bn_part as
 * in trs_gemm_f32 (per 128-row chunk).  x_dev / x_bf16_dev (optional, row stride ldx): the x0 image (fp32 for bf16 = 0,
 * bf16 for bf16 = 1) that the weight-gradient GEMM of the backward pass reads, written as a by-product by the workgroups
 * of column block 0 — bit for bit what trs_mlp_gather_concat writes.
 * Returns 0, a negative error, or 1 = shape not taken (int32 ids, B and N multiples of 256, D a multiple of 64 (bf16) /
 * 32 (fp32), at least 256 tiles of 256 x 256, 16-byte aligned operands): nothing was launched. */
int trs_mlp_gather_gemm1_fwd(const trs_tables* tables, const trs_batch* batch, int32_t passes, int32_t bf16,
                             const void* W_dev, int64_t ldw, const float* bias_dev, int64_t N, float* y_dev,
                             void* y_bf16_dev, int64_t ldy, float* bn_part_dev, float* x_dev, void* x_bf16_dev,
                             int64_t ldx, void* stream);

/* SGD on every embedding table of an MLP step, straight from d x0 = the input gradient of the first layer ((2B, ld);
 * rows [0,B) positive pass, [B,2B) negative pass; column block f = field f of x0; fp32 OR bf16 — exactly one of
 * dx0_dev / dx0_bf16_dev): W[row] -= lr * (sum of the d x0 segments of the row's references).  Replaces the sparse COO
 * gradients of the embedding tables + optimizer.step() (model.py:197-198) for the MLP scorer.  Two launches: user / item
 * rows (the user's two passes added in registers; references flagged alone in their batch — user_dup_flags_dev (B),
 * item_dup_flags_dev (B,2) from trs_epoch_flags / trs_epoch_presort, both optional — are plain read-modify-writes, the
 * rest float atomics) and the metadata tables (every workgroup owns a range of categories, sums their references in LDS
 * and applies each row once).  Needs D % 4 == 0, ld % 4 == 0 and metadata tables small enough for the owner-computes
 * form: trs_mlp_embed_sgd_update_supported(tables) != 0. */
int trs_mlp_embed_sgd_update_supported(const trs_tables* tables);
int trs_mlp_embed_sgd_update(const trs_tables* tables, const trs_batch* batch, const float* dx0_dev,
                             const void* dx0_bf16_dev, int64_t ld, float lr, const uint8_t* user_dup_flags_dev,
                             const uint8_t* item_dup_flags_dev, void* stream);

/* C(M,N) = alpha * op(A)(M,K) * op(B)(K,N) + beta * C [+ bias(N) on every row], fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain).  Row-major; transA: 0 = A stored (M,K), 1 = stored (K,M);
 * transB: 0 = B stored (K,N), 1 = stored (N,K).  Replaces aten::addmm / mm under nn.Linear and its autograd
 * (collaborative/mlp.py:107-113): forward y = x W^T (0,1), dgrad dx = dy W (0,0), wgrad dW = dy^T x (1,0; split-K
 * over workgroups with a fixed-order slab reduce).  workspace: trs_gemm_f32_workspace_bytes(M,N,K) bytes. */
int64_t trs_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int trs_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A_dev,
                 int64_t lda, const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc,
                 const float* bias_dev, float* bn_part_dev, void* workspace_dev, int64_t workspace_bytes,
                 void* stream);
/* bn_part_dev (optional, forward GEMMs): fused BatchNorm batch statistics of the output — per 128-row tile t and
 * column the mean and the sum of squared deviations from it, written to bn_part[(t*2 + {0,1})*N + col]; combine with
 * trs_bn_stats_finalize(chunk_rows = 128).  Requires beta == 0; disables split-K. */

/* Same interface and storage (fp32 in, fp32 out); the operands are rounded to bf16 (round-to-nearest-even) as they are
 * staged and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation — the `use_amp=True` mode (the reference's
 * CUDA-only fp16 autocast, model.py:86-88,280, becomes bf16 inputs / fp32 accumulate: no loss scaling needed). */
int trs_gemm_bf16(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A_dev,
                  int64_t lda, const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc,
                  const float* bias_dev, float* bn_part_dev, void* workspace_dev, int64_t workspace_bytes,
                  void* stream);

/* bf16-RESIDENT GEMM of the use_amp path (activations and weight images already stored as bf16 in HBM; fp32
 * accumulation and fp32 output; same epilogue options as trs_gemm_f32).  tn = 0: C(M,N) = alpha * A(M,K) B(N,K)^T, both
 * operands k-contiguous (forward with W, dgrad with the transposed weight image); tn = 1: C(M,N) = alpha * A(K,M)^T
 * B(K,N) (wgrad dW = dy^T x: both operands "one row per sample"; fragments come out of the transposing LDS read
 * ds_read_b64_tr_b16).  M, N multiples of 128, K a multiple of 64, 16-byte aligned rows (lda, ldb in bf16 elements,
 * multiples of 8).  Output: C_dev (fp32) or, when C_dev is NULL, C_bf16_dev (the value rounded to bf16: the forward
 * outputs y_l and the input gradients dx_l of the bf16-resident path, like autocast's half-precision linear outputs;
 * the fused BatchNorm statistics still come from the fp32 accumulators).  Replaces the reference's autocast `linear` (model.py:86-88,192-195; SURVEY 8a7: bf16 inputs, fp32
 * accumulate, no loss scaling). */
int64_t trs_gemm_bf16in_workspace_bytes(int64_t M, int64_t N, int64_t K);
int trs_gemm_bf16in(int tn, int64_t M, int64_t N, int64_t K, float alpha, const void* A_dev, int64_t lda,
                    const void* B_dev, int64_t ldb, float beta, float* C_dev, void* C_bf16_dev, int64_t ldc,
                    const float* bias_dev, float* bn_part_dev, void* workspace_dev, int64_t workspace_bytes,
                    void* stream);
/* dst (rows, cols) bf16 copy and/or dst_t (cols, rows) transposed bf16 copy of an fp32 matrix (either may be NULL):
 * the per-step refresh of the weight images the bf16-resident GEMMs read. */
int trs_f32_to_bf16(const float* src_dev, int64_t rows, int64_t cols, int64_t ld, void* dst_dev, void* dst_t_dev,
                    void* stream);
/* The same for n (1..8) matrices in ONE launch — every layer's weight images of an MLP step.  The arrays are host arrays
 * of n entries (device pointers / sizes per matrix); dst_dev[k] or dst_t_dev[k] may be NULL. */
int trs_f32_to_bf16_multi(int32_t n, const float* const* src_dev, const int64_t* rows, const int64_t* cols,
                          const int64_t* ld, void* const* dst_dev, void* const* dst_t_dev, void* stream);

/* Train-mode BatchNorm1d statistics of y (passes*rows_per_pass, H) per pass: mean_out/var_out (passes,H), biased
 * variance (chunked two-pass + Chan combination in fp64).  running_mean/var (H) non-NULL: updated once per pass in
 * pass order with `momentum` and the unbiased variance (torch.nn.BatchNorm1d under collaborative/mlp.py:82,110).
 * workspace: trs_bn_workspace_floats(rows_per_pass, H, passes) floats. */
int64_t trs_bn_workspace_floats(int64_t rows_per_pass, int32_t H, int32_t passes);
int trs_bn_batch_stats(const float* y_dev, int64_t rows_per_pass, int32_t H, int64_t ld, int32_t passes,
                       float momentum, float* mean_out_dev, float* var_out_dev, float* running_mean_dev,
                       float* running_var_dev, float* workspace_dev, void* stream);

/* Second half of trs_bn_batch_stats alone: combine chunk partials (pass, chunk, {mean, M2}, H) of `chunk_rows` rows each
 * (as trs_gemm_* writes them with chunk_rows = 128) into mean/var and update the running statistics. */
int trs_bn_stats_finalize(const float* part_dev, int64_t rows_per_pass, int32_t chunk_rows, int32_t H, int32_t passes,
                          float momentum, float* mean_out_dev, float* var_out_dev, float* running_mean_dev,
                          float* running_var_dev, void* stream);

/* out = relu(((y - mean) / sqrt(var + eps)) * gamma + beta)  (use_bn = 0: out = relu(y)); statistics indexed per
 * pass when stat_passes == passes, shared when stat_passes == 1 (eval mode: running statistics).  out_dev (fp32)
 * and/or out_bf16_dev (bf16 image for the bf16-resident GEMMs; needs H % 4 == 0), same row stride ldo in elements.
 * y_bf16 != 0: y_dev is a bf16 image (written by trs_gemm_bf16in), else fp32.
 * running_mean_dev / running_var_dev (both or neither; need use_bn and stat_passes == passes): the momentum update of
 * the running statistics from mean_dev / var_dev, pass by pass — what trs_bn_stats_finalize does when IT is given the
 * running pointers (same arithmetic, bit for bit); pass them to one of the two, not both.  Done by the forward kernel's
 * first row block, which saves the training step one launch per layer.  num_batches_tracked_dev (one int64, may be NULL;
 * with the running pointers): += passes, BatchNorm1d's batch counter, in the same place.
 * dot_w_dev (H) / dot_bias_dev (1, may be NULL) / dot_out_dev (rows_per_pass * passes): the H -> 1 output layer on the
 * last hidden layer's activations, dot_out[r] = sum_c out[r][c] * dot_w[c] + dot_bias (mlp.py:114); formed from the
 * registers of this launch when a row's columns sit in one wave (H a power of two <= 256, aligned rows) — out_dev and
 * out_bf16_dev may then both be NULL: the activations are not stored at all (trs_bn_relu_backward's outer_xw_dev gives the
 * output layer's weight gradient without them) — and by trs_rowdot on out_dev otherwise (same result up to the order of
 * the sum; an error when out_dev is NULL).
 * (collaborative/mlp.py:108-114) */
int trs_bn_relu_forward(const void* y_dev, int32_t y_bf16, int64_t rows_per_pass, int32_t passes, int32_t H, int64_t ld,
                        int32_t use_bn, int32_t stat_passes, const float* mean_dev, const float* var_dev,
                        const float* gamma_dev, const float* beta_dev, float eps, float* out_dev, void* out_bf16_dev,
                        int64_t ldo, float momentum, float* running_mean_dev, float* running_var_dev,
                        int64_t* num_batches_tracked_dev, const float* dot_w_dev, const float* dot_bias_dev,
                        float* dot_out_dev, void* stream);

/* Backward of relu(bn(y)) in train mode from dx = dL/d(out): dy (same shape), dgamma/dbeta (H) summed over both
 * passes.  use_bn = 0: dy = dx * [y > 0].  dy_colsum_dev (H, may be NULL): column sums of dy = the gradient of the
 * preceding Linear's bias, summed per pass first like trs_colsum.  dy_dev (fp32) and/or dy_bf16_dev (bf16 image for
 * the bf16-resident GEMMs; needs H % 4 == 0), row stride ldd elements.  y_bf16 / dx_bf16 != 0: that input is a bf16
 * image (row strides ld / ldd in elements).  workspace:
 * trs_bn_backward_workspace_floats(...) floats.
 * Synchronised BatchNorm under data parallelism (statistics over the GLOBAL batch): phase 1 only reduces —
 * sums_dev (passes,2,H) receives this rank's sum(d) and sum(d*xhat) per pass, dgamma / dbeta are written; the caller
 * all-reduces sums_dev (SUM) and calls phase 2, which applies with those sums and stat_rows = world * rows_per_pass as
 * the divisor.  phase 0 (sums_dev NULL, stat_rows 0) = both at once on the local batch.
 * Outer-product form (the last hidden layer: its dx is the H -> 1 output layer's input gradient g (x) w, mlp.py:115):
 * dx_dev NULL and outer_g_dev (passes*rows_per_pass) / outer_w_dev (H) given — dx[r][c] = g[r] * w[c] is formed in the
 * kernels and never stored (needs H % 4 == 0).  outer_xw_dev (H, may be NULL; outer form with BatchNorm, phase 0):
 * sum_r g[r] * relu(bn(y))[r][:] = the output layer's weight gradient (what trs_colsum(out, row_weight = g) gives on the
 * stored activations), from the activations the reduce kernel recomputes anyway. */
int64_t trs_bn_backward_workspace_floats(int64_t rows_per_pass, int32_t H, int32_t passes);
int trs_bn_relu_backward(const void* y_dev, int32_t y_bf16, const void* dx_dev, int32_t dx_bf16, int64_t rows_per_pass,
                         int32_t passes, int32_t H,
                         int64_t ld, int64_t ldd, int32_t use_bn, const float* mean_dev, const float* var_dev,
                         const float* gamma_dev, const float* beta_dev, float eps, float* dy_dev, void* dy_bf16_dev,
                         float* dgamma_dev, float* dbeta_dev, float* dy_colsum_dev, float* workspace_dev,
                         int32_t phase, float* sums_dev, int64_t stat_rows, const float* outer_g_dev,
                         const float* outer_w_dev, float* outer_xw_dev, void* stream);

/* out[h] = sum_r w[r] * x[r][h] over the passes*rows_per_pass rows (row_weight NULL: plain column sums): bias
 * gradients and the output layer's weight gradient.  Summed per pass first (identical chunking in both passes), so a
 * negative pass that is the exact negation of the positive one cancels to an exact 0 as in the reference's two
 * separate backward passes.  workspace: trs_colsum_workspace_floats(rows_per_pass, passes, H) floats. */
int64_t trs_colsum_workspace_floats(int64_t rows_per_pass, int32_t passes, int32_t H);
int trs_colsum(const float* x_dev, int64_t rows_per_pass, int32_t passes, int32_t H, int64_t ld,
               const float* row_weight_dev, float* out_dev, float* workspace_dev, void* stream);

/* Output layer H -> 1 (collaborative/mlp.py:85,113): out[r] = x[r] . w + bias[0];  and its dgrad dx[r][h] = g[r]*w[h]. */
int trs_rowdot(const float* x_dev, int64_t rows, int32_t H, int64_t ld, const float* w_dev, const float* bias_dev,
               float* out_dev, void* stream);
int trs_outer(const float* g_dev, const float* w_dev, int64_t rows, int32_t H, float* dx_dev, int64_t ld,
              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRS_H */
