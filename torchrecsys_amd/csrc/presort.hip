// presort.hip — epoch-level grouping of item references by row, and the atomic-free item update that uses it.
//
// In the dense regime (c2: 131 072 item references over 100 000 rows per step) most item rows are referenced several
// times per step, and summing those references with memory-side float atomics (~1.3 TB/s chip-wide) is the wall of the
// step.  The references of a whole epoch are known when the epoch starts (the shuffle is a keyed bijection of the
// position, the negative a counter-based draw), so they are grouped ONCE per epoch:
//   epoch_refs_kernel   every position q of the epoch -> its triple (u, i, j) [written out: K1 then reads contiguous ids
//                       instead of deriving them] and two references  key = item row, payload = (t << 1) | which;
//   batch_group_items_kernel   a counting sort in LDS, one 1024-thread workgroup per batch (the positions of a batch are
//                       contiguous already: only the row id is a key), walking the key space in chunks — no vendor
//                       library anywhere in this file since round 3 (rounds 1-2: rocprim's segmented radix sort, two 9-bit
//                       passes, 0.95 ms per 512 batches of 131 072 references against 0.64 ms for this kernel).
// Per step, sorted_item_update_kernel gives every run of equal keys to ONE lane group: it sums c * u over the run
// (c = -lr * gz of the reference, u = the still-unmodified user row) in registers and applies it with a single plain
// whole-row read-modify-write.  Runs are cut at 64-reference boundaries so a hot row (skewed data) is reduced by many
// groups in parallel; only such cut runs fall back to float atomics (one add per 64 references instead of 64).
#include <cstring>

#include "score_kernels.h"
#include "opt_rows.h"

// The plain-SGD run kernels with one float4 per lane and whole rows (D = 4*G) are compiled for 4 waves per SIMD (128 VGPRs, no spill; the
// compiler's own choice is 130 = 3 waves): random-row latency is hidden by waves in flight (c2: K2 22.0 -> 21.1 us
// between its events).  5 waves spill; the adaptive rules need 174-210 VGPRs and keep the default.
#ifndef K2_WAVES
#define K2_WAVES 4
#endif
namespace trs {

struct RefPayload {
  uint32_t tw;  // (t << 1) | which   (t = position inside the batch, which: 0 positive / 1 negative)
};

struct EpochArgs {
  const int2* sui;  // resident stream {user, item} or NULL (ids given in user/pos/neg)
  const int32_t* neg_static;
  int64_t N;
  uint64_t shuffle_key;
  int hb;
  uint64_t sample_seed;
  int64_t first_pos;  // epoch position of the first triple
  int64_t n_pos;      // triples covered (whole batches)
  int64_t batch;
  int64_t n_users, n_items;
  int item_bits;
  int32_t* user;  // (n_pos) in/out
  int32_t* pos;
  int32_t* neg;
  void* keys;  // (2*n_pos) uint32 or uint64
  RefPayload* vals;
  int32_t* err;
  TrsSampler S;  // sampler options (max_tries == 0: the reference's sampler); k_neg: the epoch has N * k_neg positions
};

template <typename KeyT, int SRC>
__global__ __launch_bounds__(TRS_BLOCK) void epoch_refs_kernel(const EpochArgs a) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  KeyT* keys = reinterpret_cast<KeyT*>(a.keys);
  for (int64_t q = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; q < a.n_pos; q += stride) {
    int64_t u, i, j;
    if (SRC != 0) {
      // (k_neg > 1: N * k_neg epoch positions, every stream row visited k_neg times, each time with its own negative)
      const int64_t p = trs_feistel_perm(a.first_pos + q, a.N * a.S.k_neg, a.shuffle_key, a.hb) % a.N;
      const int2 ui = a.sui[p];
      u = ui.x;
      i = ui.y;
      j = SRC == 2 ? (int64_t)a.neg_static[p]
                   : ((uint64_t)u < (uint64_t)a.n_users && (uint64_t)i < (uint64_t)a.n_items
                          ? trs_sample_neg_opt(a.sample_seed, (uint64_t)(a.first_pos + q), u, i, a.n_items, a.S)
                          : 0);
    } else {
      u = a.user[q];
      i = a.pos[q];
      j = a.neg[q];
    }
    bool ok = true;
    if ((uint64_t)u >= (uint64_t)a.n_users) { ok = false; u = 0; }
    if ((uint64_t)i >= (uint64_t)a.n_items) { ok = false; i = 0; }
    if ((uint64_t)j >= (uint64_t)a.n_items) { ok = false; j = 0; }
    if (!ok && a.err) atomicOr(a.err, 1);  // reported as IndexError at the end of the epoch
    if (SRC != 0 || !ok) {
      a.user[q] = (int32_t)u;
      a.pos[q] = (int32_t)i;
      a.neg[q] = (int32_t)j;
    }
    if (keys) {  // (NULL: ids only — the sparse regime's flags need no sort keys)
      keys[2 * q] = (KeyT)i;
      keys[2 * q + 1] = (KeyT)j;
    }
  }
}

// ---------------------------------------------------------------------------------------------- duplicate flags only
// The sparse regime (c4: 65 536 item references over 1M rows, 32 768 users over 10M per step) needs no grouping at all:
// 94 % of the item references and 99.7 % of the users are alone on their row in the batch, K1 updates those rows in
// place, and the few shared rows are updated by K2's float atomics (fast_step.hip, INL 2).  All K1 needs from the epoch is
// one flag per reference: "another reference of this batch names the same row".  One 1024-thread workgroup per batch
// finds them with a bitmap in LDS (2^20 bits = 128 KB of the CU's 160 KB):
//   A  every reference sets its row's bit (ds_or returning the old word): a bit already set = a LATER arrival;
//   B  bitmap cleared; the later arrivals set their row's bit again: the bitmap is now the set of shared rows;
//   C  every reference reads its row's bit = its flag.
// Tables with more rows than bits are hashed into the bitmap (multiplicative hash): a collision only raises a flag for a
// row that is in fact alone, and a flagged lone row is still updated exactly (through the accumulator) — the flags are
// conservative, never wrong.  The same kernel writes the batch's ids first (what epoch_refs_kernel does), so the whole
// presort of the sparse regime is this one launch: no sort, no vendor library.
struct FlagArgs {
  const int2* sui;  // resident stream {user, item} or NULL (ids given in user/pos/neg)
  const int32_t* neg_static;
  int64_t N;
  uint64_t shuffle_key;
  int hb;
  uint64_t sample_seed;
  int64_t first_pos, batch;
  int64_t n_users, n_items;
  int32_t *user, *pos, *neg;  // (n_batches * batch) in/out
  uint8_t* uflags;            // (n_batches * batch)
  uint8_t* iflags;            // (n_batches * batch, 2)
  int32_t* err;
  int log2_bits;              // bitmap size
  int item_hash, user_hash;   // 0: row id = bit (table fits the bitmap), 1: hashed
  int what_first, what_end;   // tables flagged: [0, 2) items then users; [1, 2) users only (the dense regime sorts its items)
  int32_t* nflag;             // NULL, or (n_batches): phase D puts the triples with a flagged reference first and counts them
  int ord_off, ord_cap;       // phase D's LDS scratch: words in front of the two position lists, entries per list
};

constexpr int FLAG_THREADS = 1024;

__device__ __forceinline__ uint32_t flag_bit(int32_t row, int hashed, int log2_bits) {
  return hashed ? ((uint32_t)row * 2654435761u) >> (32 - log2_bits) : (uint32_t)row;
}

// Every pass walks the batch FLAG_U triples per thread at a time with all their global loads issued before the first
// LDS operation: the kernel is one workgroup per batch, i.e. a chain of load latencies, not bandwidth.  A thread owns
// the same triples in every pass (t = tid + k * 1024), so what phase A learns ("my reference came later") stays in
// registers — one bit per reference — until phase B needs it; only the final flags go to memory.
constexpr int FLAG_U = 8;
constexpr int FLAG_ROUNDS = 8;       // rounds per 64-bit word of "came later" bits: 8 rounds x FLAG_U triples
constexpr int FLAG_MAX_GROUPS = 4;   // batch <= 1024 * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS = 262 144 triples

// NG: 64-bit words of "came later" bits a thread keeps per id stream (compile-time: the words stay in registers).  The
// batch's ids are given (trs_epoch_flags writes them first with a wide launch of epoch_refs_kernel).
template <int NG>
__global__ __launch_bounds__(FLAG_THREADS) void batch_flags_kernel(const FlagArgs a) {
  extern __shared__ uint32_t bm[];
  const int words = 1 << (a.log2_bits - 5);
  const int64_t q0 = (int64_t)blockIdx.x * a.batch;
  const int B = (int)a.batch;
  const int lb = a.log2_bits;
  uint16_t* iflags16 = reinterpret_cast<uint16_t*>(a.iflags);  // {pos, neg} of a position as one 16-bit word
  constexpr int STEP = FLAG_THREADS * FLAG_U;
  uint64_t anyf[NG];  // bit (rd % 8) * FLAG_U + k of word rd / 8: triple (round rd, k) of this thread carries a flag (phase D)
#pragma unroll
  for (int g = 0; g < NG; ++g) anyf[g] = 0;
  // ---- item references, then users: phases A / B / C over the same LDS bitmap
  for (int what = a.what_first; what < a.what_end; ++what) {
    const int hashed = what == 0 ? a.item_hash : a.user_hash;
    const int32_t* ida = what == 0 ? a.pos : a.user;
    // bit (rd % 8) * FLAG_U + k of word rd / 8: the first / second id of triple (round rd, k) came later
    uint64_t lat0[NG], lat1[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) lat0[g] = lat1[g] = 0;
    for (int w = threadIdx.x; w < words; w += FLAG_THREADS) bm[w] = 0u;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll 1
    for (int rr = 0; rr < FLAG_ROUNDS && (g * FLAG_ROUNDS + rr) * STEP < B; ++rr) {  // A: set; a bit already set = a later arrival
      const int rd = g * FLAG_ROUNDS + rr;
      const int t0 = threadIdx.x + rd * STEP;
      uint32_t b0[FLAG_U], b1[FLAG_U];
#pragma unroll
      for (int k = 0; k < FLAG_U; ++k) {
        const int t = t0 + k * FLAG_THREADS;
        const int64_t q = q0 + (t < B ? t : 0);
        b0[k] = flag_bit(ida[q], hashed, lb);
        b1[k] = what == 0 ? flag_bit(a.neg[q], hashed, lb) : 0u;
      }
#pragma unroll
      for (int k = 0; k < FLAG_U; ++k) {
        if (t0 + k * FLAG_THREADS < B) {
          const uint32_t o0 = atomicOr(&bm[b0[k] >> 5], 1u << (b0[k] & 31));
          lat0[g] |= (uint64_t)((o0 >> (b0[k] & 31)) & 1u) << (rr * FLAG_U + k);
          if (what == 0) {
            const uint32_t o1 = atomicOr(&bm[b1[k] >> 5], 1u << (b1[k] & 31));
            lat1[g] |= (uint64_t)((o1 >> (b1[k] & 31)) & 1u) << (rr * FLAG_U + k);
          }
        }
      }
    }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += FLAG_THREADS) bm[w] = 0u;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {  // B: the later arrivals mark their row as shared (few: ids re-read)
      for (uint64_t m = lat0[g]; m; m &= m - 1) {
        const int bit = __ffsll((unsigned long long)m) - 1;
        const int64_t q = q0 + threadIdx.x + (int64_t)(g * FLAG_ROUNDS + bit / FLAG_U) * STEP + (bit % FLAG_U) * FLAG_THREADS;
        const uint32_t b = flag_bit(ida[q], hashed, lb);
        atomicOr(&bm[b >> 5], 1u << (b & 31));
      }
      for (uint64_t m = lat1[g]; m; m &= m - 1) {
        const int bit = __ffsll((unsigned long long)m) - 1;
        const int64_t q = q0 + threadIdx.x + (int64_t)(g * FLAG_ROUNDS + bit / FLAG_U) * STEP + (bit % FLAG_U) * FLAG_THREADS;
        const uint32_t b = flag_bit(a.neg[q], hashed, lb);
        atomicOr(&bm[b >> 5], 1u << (b & 31));
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll 1
    for (int rr = 0; rr < FLAG_ROUNDS && (g * FLAG_ROUNDS + rr) * STEP < B; ++rr) {  // C: flag = "my row is shared"
      const int t0 = threadIdx.x + (g * FLAG_ROUNDS + rr) * STEP;
      uint32_t b0[FLAG_U], b1[FLAG_U];
#pragma unroll
      for (int k = 0; k < FLAG_U; ++k) {
        const int t = t0 + k * FLAG_THREADS;
        const int64_t q = q0 + (t < B ? t : 0);
        b0[k] = flag_bit(ida[q], hashed, lb);
        b1[k] = what == 0 ? flag_bit(a.neg[q], hashed, lb) : 0u;
      }
#pragma unroll
      for (int k = 0; k < FLAG_U; ++k) {
        const int t = t0 + k * FLAG_THREADS;
        if (t < B) {
          const uint32_t s0 = (bm[b0[k] >> 5] >> (b0[k] & 31)) & 1u;
          const uint32_t s1 = what == 0 ? (bm[b1[k] >> 5] >> (b1[k] & 31)) & 1u : 0u;
          if (what == 0) iflags16[q0 + t] = (uint16_t)(s0 | (s1 << 8));
          else a.uflags[q0 + t] = (uint8_t)s0;
          anyf[g] |= (uint64_t)(s0 | s1) << (rr * FLAG_U + k);
        }
      }
    }
    }
    __syncthreads();
  }
  if (!a.nflag) return;
  // ---- D: flagged-first order.  The one-launch step (fwd_stage_kernel INL 3) may add a flagged reference's update into
  // its table only when NO workgroup reads that row any more; if the triples that carry a flagged reference are the
  // batch's first nf, every workgroup is past them after its first ceil(nf / stride) iterations and counts itself in
  // there — the grid-wide wait at the end of the launch finds the count complete.  The order inside a batch means
  // nothing to the step (a sum over the batch), so the batch is partitioned in place: nf = the number of flagged triples;
  // the k-th NON-flagged triple at a position < nf (a hole) changes places with the k-th flagged triple at a position >=
  // nf (a mover) — ranks from per-64 ballots and one scan, positions through two LDS lists (the bitmap is dead by now),
  // then m disjoint swaps of (user, pos, neg, flags).  Deterministic; a batch with more misplaced triples than the lists
  // hold is left as it is and reported as nf = batch (the step then counts in after its last iteration, as before).
  {
    const int cells = (B + 63) >> 6;
    uint32_t* F = bm;  // flagged triples in front of cell c (cell = 64 consecutive positions)
    uint32_t* holes = bm + a.ord_off;
    uint32_t* movers = holes + a.ord_cap;
    __shared__ uint32_t s_nf, s_flnf;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NWV = FLAG_THREADS / 64;
    // a thread's triples (t = tid + j * 1024, j = rd * FLAG_U + k) are lane `lane` of the cells wv + 16 j, and their flags
    // sit in its registers (anyf): the three passes below touch no global memory
#define TRS_FOR_MY_CELLS(BODY)                                                                     \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                \
      _Pragma("unroll 1") for (int rr = 0; rr < FLAG_ROUNDS && (g * FLAG_ROUNDS + rr) * STEP < B; ++rr) { \
        _Pragma("unroll") for (int k = 0; k < FLAG_U; ++k) {                                        \
          const int j = (g * FLAG_ROUNDS + rr) * FLAG_U + k;                                       \
          const int c = wv + NWV * j;                                                              \
          const int t = c * 64 + lane;                                                             \
          const bool f = t < B && ((anyf[g] >> (rr * FLAG_U + k)) & 1ull) != 0;                    \
          const uint64_t bal = __ballot(f);                                                        \
          if (c < cells) { BODY }                                                                  \
        }                                                                                          \
      }                                                                                            \
    }
    TRS_FOR_MY_CELLS(if (lane == 0) F[c] = (uint32_t)__popcll((unsigned long long)bal);)
    __syncthreads();
    if (wv == 0) {  // exclusive scan over the cells: a lane sums `per` consecutive cells, the wave scans the 64 sums
      const int per = (cells + 63) >> 6;
      uint32_t sum = 0;
      for (int i = 0; i < per; ++i) {
        const int c = lane * per + i;
        if (c < cells) sum += F[c];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      uint32_t run = incl - sum;
      for (int i = 0; i < per; ++i) {
        const int c = lane * per + i;
        if (c < cells) {
          const uint32_t v = F[c];
          F[c] = run;
          run += v;
        }
      }
      if (lane == 63) s_nf = s_flnf = incl;  // (s_flnf: flagged triples in front of position nf; nf = batch has no such position)
    }
    __syncthreads();
    const int nf = (int)s_nf;
    TRS_FOR_MY_CELLS(
      const uint32_t fl = F[c] + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
      if (t == nf) s_flnf = fl;
      if (t < nf && !f && t - fl < (uint32_t)a.ord_cap) holes[t - fl] = (uint32_t)t;)
    __syncthreads();
    const uint32_t flnf = s_flnf;
    const int m = nf - (int)flnf;  // holes = movers
    if (m > a.ord_cap) {
      if (threadIdx.x == 0) a.nflag[blockIdx.x] = B;
      return;
    }
    TRS_FOR_MY_CELLS(
      const uint32_t fl = F[c] + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
      if (t >= nf && f) movers[fl - flnf] = (uint32_t)t;)
#undef TRS_FOR_MY_CELLS
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += FLAG_THREADS) {
      const int64_t x = q0 + holes[k], y = q0 + movers[k];
      const int32_t ux = a.user[x], px = a.pos[x], nx = a.neg[x], uy = a.user[y], py = a.pos[y], ny = a.neg[y];
      const uint8_t fux = a.uflags[x], fuy = a.uflags[y];
      const uint16_t fix = iflags16[x], fiy = iflags16[y];
      a.user[x] = uy; a.pos[x] = py; a.neg[x] = ny; a.uflags[x] = fuy; iflags16[x] = fiy;
      a.user[y] = ux; a.pos[y] = px; a.neg[y] = nx; a.uflags[y] = fux; iflags16[y] = fix;
    }
    if (threadIdx.x == 0) a.nflag[blockIdx.x] = nf;
  }
}

// ---------------------------------------------------------------------------------------------- grouping by item row
// The dense regime's sorted runs need every batch's 2B item references grouped by row.  A segmented radix sort moves
// keys AND payloads through global memory twice (two 9-bit passes: 32 B per reference); a counting sort reads the ids
// (L2-resident: one batch = 8 B per triple) and writes each (row, payload) pair once.  One 1024-thread workgroup per
// batch walks the key space in chunks of GRP_BINS rows; per chunk, in LDS, GRP_BINS cursors (64 KB) and a staging buffer
// of GRP_CAP packed words (88 KB):
//   count    every reference whose row falls in the chunk: ds_add on its counter (ids read straight from pos / neg);
//   scan     counters -> first output slot of every row (each wave scans a contiguous segment 64 rows at a time, wave
//            totals combined through LDS); the same sweep finds the rows at which the output crosses a multiple of
//            GRP_CAP = the windows of rows whose output fits the staging buffer (one window unless rows are hot);
//   scatter  per window, every reference again: slot = ds_add_rtn on its row's cursor; (row in chunk, payload) packed
//            into one word of the staging buffer at slot - window start — NOT stored to global memory: scattered 4-byte
//            stores cost 1.8 ms per 512 batches of 65 536 (L2 request rate; tools/micro/group_bench.hip), the staged
//            ones 0.15 ms;
//   write    the staging buffer leaves as coalesced stores of keys and payloads.
// A row with more references than the buffer holds (hot rows of skewed data) spills its tail as direct stores.
// Output = the batch's references in ascending row order (inside a row: arrival order — it only permutes the fp32
// summation order of the run).  Measured at c2 (100 K rows = 7 chunks, 14 sweeps): 0.64 ms per 512 batches against
// 0.88 ms for rocprim's segmented radix sort on the same keys; 20 K rows: 0.19 against 0.41 ms.
// Any table size (the chunk loop just gets longer: every batch of the slice is grouped by its own workgroup at the same
// time, so a 1M-row table's 61 chunks cost the slice 61 x 2 sweeps over ids that sit in L2, not 61 x the slice) and any
// batch up to 2^21 references: a staged word packs (row in chunk, payload) into 32 bits, so a batch whose payload needs
// more than 18 bits takes chunks of 2^(32 - payload bits) rows instead of 2^14 (BIN_BITS template parameter).
constexpr int GRP_THREADS = 1024, GRP_MAX_BIN_BITS = 14, GRP_MIN_BIN_BITS = 10, GRP_CAP = 22528, GRP_U = 8;
constexpr int GRP_LDS_WORDS = (1 << GRP_MAX_BIN_BITS) + GRP_CAP;  // cursors + staging buffer (152 KB)
constexpr int GRP_WMAX = 256;  // windows per chunk: 2B / GRP_CAP + 2

// SRC 0: the references' rows are the pos / neg ids (payload 2t + w).  SRC 1: they come interleaved, keys[2t] / keys[2t+1]
// (the metadata columns: `pos_all` is that array, read 8 bytes per triple).  SRC 2: ONE reference per position — the
// users of a batch (`pos_all` = user ids): B references per batch, payload = the position in the SLICE, b * batch + t.
template <int SRC, int BIN_BITS>
__global__ __launch_bounds__(GRP_THREADS) void batch_group_items_kernel(const int32_t* __restrict__ pos_all,
                                                                       const int32_t* __restrict__ neg_all,
                                                                       int64_t batch, int64_t n_items, int pb,
                                                                       uint32_t* __restrict__ keys_all,
                                                                       RefPayload* __restrict__ vals_all) {
  constexpr int GRP_BINS = 1 << BIN_BITS;
  extern __shared__ uint32_t grp_lds[];
  uint32_t* cur = grp_lds;               // GRP_BINS counters, then cursors
  uint32_t* stage = grp_lds + GRP_BINS;  // GRP_CAP packed (row in chunk << pb) | payload
  __shared__ uint32_t wave_tot[GRP_THREADS / TRS_WAVE];
  __shared__ uint32_t win_row[GRP_WMAX + 2], win_slot[GRP_WMAX + 2];  // first row / first slot of every window
  __shared__ uint32_t tot_s;
  const int32_t* pos = pos_all + (SRC == 1 ? 2 : 1) * (int64_t)blockIdx.x * batch;
  const int32_t* neg = SRC != 0 ? nullptr : neg_all + (int64_t)blockIdx.x * batch;
  const uint2* pair = reinterpret_cast<const uint2*>(pos);
  constexpr int PER = SRC == 2 ? 1 : 2;  // references per position
  uint32_t* keys = keys_all + PER * (int64_t)blockIdx.x * batch;
  RefPayload* vals = vals_all + PER * (int64_t)blockIdx.x * batch;
  const uint32_t pay0 = SRC == 2 ? (uint32_t)((int64_t)blockIdx.x * batch) : 0u;  // added to the payload on the way out
  const int B = (int)batch;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int NW = GRP_THREADS / TRS_WAVE, SEG = GRP_BINS / NW;  // rows per wave in the scan
  constexpr int STEP = GRP_THREADS * GRP_U;
  const uint32_t pmask = (1u << pb) - 1u;
  uint32_t base = 0;  // output slot of the chunk's first reference
  for (int64_t c0l = 0; c0l < n_items; c0l += GRP_BINS) {
    const uint32_t c0 = (uint32_t)c0l;
    for (int w = threadIdx.x; w < GRP_BINS; w += GRP_THREADS) cur[w] = 0u;
    __syncthreads();
    for (int t0 = threadIdx.x; t0 < B; t0 += STEP) {  // count
      uint32_t kp[GRP_U], kn[GRP_U];
#pragma unroll
      for (int k = 0; k < GRP_U; ++k) {
        const int t = t0 + k * GRP_THREADS;
        if (SRC == 1) {
          const uint2 kk = pair[t < B ? t : 0];
          kp[k] = kk.x - c0;
          kn[k] = kk.y - c0;
        } else if (SRC == 2) {
          kp[k] = (uint32_t)pos[t < B ? t : 0] - c0;
          kn[k] = 0xFFFFFFFFu;  // (never in a chunk)
        } else {
          kp[k] = (uint32_t)pos[t < B ? t : 0] - c0;
          kn[k] = (uint32_t)neg[t < B ? t : 0] - c0;
        }
      }
#pragma unroll
      for (int k = 0; k < GRP_U; ++k) {
        if (t0 + k * GRP_THREADS < B) {
          if (kp[k] < (uint32_t)GRP_BINS) atomicAdd(&cur[kp[k]], 1u);
          if (kn[k] < (uint32_t)GRP_BINS) atomicAdd(&cur[kn[k]], 1u);
        }
      }
    }
    __syncthreads();
    // exclusive scan (slots relative to the chunk's first): wave wv owns rows [wv * SEG, (wv + 1) * SEG)
    uint32_t tot = 0;
    for (int i = lane; i < SEG; i += TRS_WAVE) tot += cur[wv * SEG + i];
    tot = (uint32_t)trs_wave_sum_i((int)tot);
    if (lane == 0) wave_tot[wv] = tot;
    __syncthreads();
    uint32_t carry = 0;
    for (int w = 0; w < wv; ++w) carry += wave_tot[w];
    for (int i0 = 0; i0 < SEG; i0 += TRS_WAVE) {
      const uint32_t r = (uint32_t)(wv * SEG + i0 + lane);
      const uint32_t v = cur[r];
      uint32_t inc = v;  // inclusive scan over the 64 lanes
#pragma unroll
      for (int o = 1; o < TRS_WAVE; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += up;
      }
      const uint32_t f = carry + inc - v, e = f + v;  // this row's slots [f, e)
      cur[r] = f;
      // the row at which the output crosses w * GRP_CAP closes window w - 1 (a hot row may close several)
      for (uint32_t w = f / GRP_CAP + 1; w * GRP_CAP <= e; ++w) {
        win_row[w] = r + 1;
        win_slot[w] = e;
      }
      carry += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (threadIdx.x == GRP_THREADS - 1) {  // (last wave: carry = the chunk's references)
      const uint32_t nw = carry / GRP_CAP + 1;
      tot_s = carry;
      win_row[0] = 0u;
      win_slot[0] = 0u;
      win_row[nw] = (uint32_t)GRP_BINS;
      win_slot[nw] = carry;
    }
    __syncthreads();
    const uint32_t totc = tot_s, nwin = totc / GRP_CAP + 1;
    for (uint32_t w = 0; w < nwin; ++w) {
      const uint32_t r0 = win_row[w], nr = win_row[w + 1] - r0, wb = win_slot[w], wn = win_slot[w + 1] - wb;
      if (nr == 0u || wn == 0u) continue;  // (uniform: LDS values)
      for (int t0 = threadIdx.x; t0 < B; t0 += STEP) {  // scatter into the staging buffer
        uint32_t kp[GRP_U], kn[GRP_U];
#pragma unroll
        for (int k = 0; k < GRP_U; ++k) {
          const int t = t0 + k * GRP_THREADS;
          if (SRC == 1) {
            const uint2 kk = pair[t < B ? t : 0];
            kp[k] = kk.x - c0;
            kn[k] = kk.y - c0;
          } else if (SRC == 2) {
            kp[k] = (uint32_t)pos[t < B ? t : 0] - c0;
            kn[k] = 0xFFFFFFFFu;  // (never in a chunk)
          } else {
            kp[k] = (uint32_t)pos[t < B ? t : 0] - c0;
            kn[k] = (uint32_t)neg[t < B ? t : 0] - c0;
          }
        }
#pragma unroll
        for (int k = 0; k < GRP_U; ++k) {
          const int t = t0 + k * GRP_THREADS;
          if (t < B) {
            if (kp[k] - r0 < nr) {
              const uint32_t slot = atomicAdd(&cur[kp[k]], 1u), o = slot - wb;
              if (o < (uint32_t)GRP_CAP) stage[o] = (kp[k] << pb) | ((uint32_t)PER * (uint32_t)t);
              else { keys[base + slot] = kp[k] + c0; vals[base + slot].tw = (uint32_t)PER * (uint32_t)t + pay0; }
            }
            if (kn[k] - r0 < nr) {
              const uint32_t slot = atomicAdd(&cur[kn[k]], 1u), o = slot - wb;
              if (o < (uint32_t)GRP_CAP) stage[o] = (kn[k] << pb) | (2u * (uint32_t)t + 1u);
              else { keys[base + slot] = kn[k] + c0; vals[base + slot].tw = 2u * (uint32_t)t + 1u + pay0; }
            }
          }
        }
      }
      __syncthreads();
      const uint32_t nst = wn < (uint32_t)GRP_CAP ? wn : (uint32_t)GRP_CAP;
      for (uint32_t o = threadIdx.x; o < nst; o += GRP_THREADS) {  // coalesced write-out
        const uint32_t word = stage[o];
        keys[base + wb + o] = (word >> pb) + c0;
        vals[base + wb + o].tw = (word & pmask) + pay0;
      }
      __syncthreads();
    }
    base += totc;
  }
}

// ---------------------------------------------------------------------------------------------- per-step update
struct SortedArgs {
  trs_tables T;
  const void* keys;         // this step's 2B sorted keys
  const RefPayload* vals;   // this step's 2B sorted payloads
  const int32_t* user_ids;  // this step's B user ids (position t -> user): the unstaged form reads the user table
  int64_t B;
  int item_bits;
  const float* gz;          // (2,B)
  float lr;
  uint64_t* uown;       // NULL: the user-duplicate stamps are not needed (flags were precomputed)
  uint32_t* udup;
  uint32_t stamp;
  const float* ustage;  // (B,D) pre-update user rows staged by K1 (indexed by t), or NULL: read the user table
  OptArgs o;            // staged form: update rule (OPT_SGD: lr above); adaptive rules coalesce whole runs first
  int parity;           // step parity: which cut-run counter this step appends to
  // metadata scorers (staged form, SGD): the staged rows are X(t, which) = ustage[which * xpass + t]; FM with metadata
  // stages the per-pass field sums S (xpass = B) and the run applies w += sum(c*S) - sum(c)*w (gradient g*(S - w) of
  // every reference, w = the pre-update row); Linear / FM without metadata stage the user row (xpass = 0, fmsub = 0)
  int64_t xpass;
  int fmsub;
  // the table the runs update: the item table (NULL: T.item / T.item_lin) or one metadata column's tables
  float* tab;
  float* tab_lin;  // never NULL with tab (a Linear scorer's metadata columns pass a scratch array)
  // 1: K1 has already applied the update of every reference that is alone on its row (item-duplicate flags,
  // item_flags_kernel): runs of length one are not walked here
  int skip_single;
};

constexpr int RUN_CHUNK = 64;  // runs are cut at multiples of this many references

// STAGED: user rows come from K1's staging buffer (row t of the batch; K1 has already updated the table rows of users
// referenced once) and no duplicate stamping is needed.
template <typename KeyT, int VEC, int G, int K, bool FULL, bool STAGED>
__global__ __launch_bounds__(TRS_BLOCK) void sorted_item_update_kernel(const SortedArgs a) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t n = 2 * a.B;
  const KeyT* keys = reinterpret_cast<const KeyT*>(a.keys);
  const KeyT row_mask = (KeyT)(((uint64_t)1 << a.item_bits) - 1);
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  const uint64_t hi = (uint64_t)a.stamp << 32;
  const int64_t niter = (n + TPW - 1) / TPW;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t i = it * TPW + lane / G;
    const bool valid = i < n;
    const int64_t ic = valid ? i : n - 1;
    // everything the first element of a run needs is requested up front, unconditionally (a conditional block that
    // contains a load would end in s_waitcnt vmcnt(0)); groups that turn out not to lead a run read row 0 instead
    const KeyT key = keys[ic];
    const KeyT prev = keys[ic > 0 ? ic - 1 : 0];
    const RefPayload me = a.vals[ic];
    const int64_t me_user = STAGED ? 0 : (int64_t)a.user_ids[me.tw >> 1];
    const bool head = ic == 0 || prev != key;
    const bool leader = valid && (head || (ic % RUN_CHUNK) == 0);
    uint64_t own = 0;
    if (!STAGED) own = a.uown[me_user];
    const float* ubase = STAGED ? a.ustage : T.user;
    const float c0 = -a.lr * a.gz[(int64_t)(me.tw & 1u) * a.B + (me.tw >> 1)];
    const int64_t row = (int64_t)(key & row_mask);
    RowReg<VEC, K> u0, w;
    row_load<VEC, G, K, FULL>(u0, ubase, leader ? (STAGED ? (int64_t)(me.tw >> 1) : me_user) : 0, D, lig);
    row_load<VEC, G, K, FULL>(w, T.item, leader ? row : 0, D, lig);
    const float wl = T.item_lin[leader ? row : 0];
    // the positive reference of a triple also checks whether its user row has other references in this step (K3)
    if (!STAGED && valid && (me.tw & 1u) == 0 && lig == 0 && own != (hi | (uint64_t)(me.tw >> 1)))
      a.udup[me_user] = a.stamp;
    if (!leader) continue;
    const int64_t chunk_end = (ic / RUN_CHUNK + 1) * RUN_CHUNK < n ? (ic / RUN_CHUNK + 1) * RUN_CHUNK : n;
    RowReg<VEC, K> acc;
#pragma unroll
    for (int q = 0; q < N; ++q) acc.v[q] = c0 * u0.v[q];
    float lin = c0;
    int64_t j = ic + 1;
    while (j < chunk_end && keys[j] == key) {  // further references of the same row (short: 1.3 per row at c2)
      const RefPayload pl = a.vals[j];
      const float c = -a.lr * a.gz[(int64_t)(pl.tw & 1u) * a.B + (pl.tw >> 1)];
      RowReg<VEC, K> u;
      row_load<VEC, G, K, FULL>(u, ubase, STAGED ? (int64_t)(pl.tw >> 1) : (int64_t)a.user_ids[pl.tw >> 1], D, lig);
#pragma unroll
      for (int q = 0; q < N; ++q) acc.v[q] += c * u.v[q];
      lin += c;
      ++j;
    }
    // the run is the whole segment iff it starts at a head and did not stop at a chunk boundary inside the segment
    const bool cut_tail = j == chunk_end && j < n && keys[j] == key;
    float* irow = T.item + row * (int64_t)D;
    if (head && !cut_tail) {
#pragma unroll
      for (int q = 0; q < N; ++q) w.v[q] += acc.v[q];
      row_store<VEC, G, K>(w, irow, D, lig);
      if (lig == 0) T.item_lin[row] = wl + lin;
    } else {  // a cut piece of a long segment (hot row): several groups add into the row
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const int e = (kk * G + lig) * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q)
          if (e + q < D) atomicAdd(irow + e + q, acc.v[kk * VEC + q]);
      }
      if (lig == 0) atomicAdd(T.item_lin + row, lin);
    }
  }
}

// STAGED form, wave-cooperative: a 256-thread workgroup owns one 64-reference chunk (= RUN_CHUNK, so a run never leaves
// the chunk).  Every wave detects the run leaders with ONE lane per reference (coalesced keys / payloads, then the
// reference's coefficient c = -lr*gz), ranks them (ballot + popcount) and publishes rank -> lane through LDS; the
// leaders are then dealt round-robin to the lane groups of the four waves.  A group knows its run's members from the
// continuation mask, gets their staging rows t and coefficients by shuffle, and issues the item row plus up to MEMB
// member rows for SLOTS runs at once — no load sits under a branch and no load depends on a loop-carried compare.
constexpr int SI_SLOTS = 3, SI_MEMB = 3;

template <typename KeyT, int VEC, int G, int K, bool FULL, int OPT = OPT_SGD>
__device__ __forceinline__ void sorted_item_update_staged_body(const SortedArgs& a, int block_id, int n_blocks) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  constexpr int NW = TRS_BLOCK / TRS_WAVE;
  __shared__ unsigned char lead_lane[NW][TRS_WAVE];
  const trs_tables& T = a.T;
  const int D = T.D;
  float* const tab = a.tab ? a.tab : T.item;
  float* const tab_lin = a.tab ? a.tab_lin : T.item_lin;
  const int64_t n = 2 * a.B;
  const KeyT* keys = reinterpret_cast<const KeyT*>(a.keys);
  const KeyT row_mask = (KeyT)(((uint64_t)1 << a.item_bits) - 1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lig = lane % G, gi = lane / G;
  const int64_t nchunk = (n + RUN_CHUNK - 1) / RUN_CHUNK;
  for (int64_t ch = block_id; ch < nchunk; ch += n_blocks) {
    const int64_t base = ch * RUN_CHUNK;
    const int64_t i = base + lane;
    const bool valid = i < n;
    const int64_t il = valid ? i : n - 1;
    const KeyT k0 = keys[il];
    const KeyT kp = keys[il > 0 ? il - 1 : 0];
    const KeyT knext = keys[base + RUN_CHUNK < n ? base + RUN_CHUNK : n - 1];  // first key of the next chunk
    const RefPayload me = a.vals[il];
    const int t = (int)(me.tw >> 1);
    const int xrow = t + (int)(me.tw & 1u) * (int)a.xpass;  // row of the staged image this reference multiplies
    // SGD is linear in the gradient: the learning rate is folded into the coefficient.  The adaptive rules need the
    // coalesced gradient itself: c = gz.
    const float gzv = a.gz[(int64_t)(me.tw & 1u) * a.B + t];
    const float c = OPT == OPT_SGD ? -a.lr * gzv : gzv;
    const bool head = il == 0 || kp != k0;                // first reference of the row in this step
    const bool cont = valid && lane > 0 && !head;         // continues the run of the lane before it
    // a reference alone on its row (same test as item_flags_kernel): K1 has updated that row already
    const KeyT kn1 = keys[il + 1 < n ? il + 1 : il];
    const bool single = head && !(il + 1 < n && kn1 == k0);
    const bool lead = valid && !cont && !(a.skip_single != 0 && single);
    const uint64_t lmask = __ballot(lead), cmask = __ballot(cont);
    const int nlead = __popcll(lmask);
    const int rank = __popcll(lmask & (((uint64_t)1 << lane) - 1));
    if (lead) lead_lane[wv][rank] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t klo = (uint32_t)(k0 & (KeyT)0xffffffffu), khi = (uint32_t)((uint64_t)k0 >> 32);
    // rounds of TPW leaders; round r belongs to wave r % NW; SLOTS rounds per pass
    for (int r0 = wv; r0 * TPW < nlead; r0 += NW * SI_SLOTS) {
      int l[SI_SLOTS], len[SI_SLOTS], hd[SI_SLOTS];
      KeyT keyv[SI_SLOTS];
      bool has[SI_SLOTS];
      int64_t row[SI_SLOTS];
      RowReg<VEC, K> w[SI_SLOTS], u[SI_SLOTS][SI_MEMB];
      RowReg<VEC, K> s1[SI_SLOTS], s2[SI_SLOTS];  // optimiser state rows (dead for SGD)
      float wl[SI_SLOTS], cm[SI_SLOTS][SI_MEMB], ls1[SI_SLOTS], ls2[SI_SLOTS];
#pragma unroll
      for (int s = 0; s < SI_SLOTS; ++s) {
        const int want = (r0 + s * NW) * TPW + gi;
        has[s] = want < nlead;
        l[s] = has[s] ? (int)lead_lane[wv][want] : 0;
        const uint64_t after = l[s] < 63 ? (cmask >> (l[s] + 1)) : 0ull;
        len[s] = 1 + (int)__ffsll((unsigned long long)~after) - 1;
        const uint64_t key = ((uint64_t)(uint32_t)__shfl((int)khi, l[s], 64) << 32) | (uint32_t)__shfl((int)klo, l[s], 64);
        keyv[s] = (KeyT)key;
        hd[s] = __shfl((int)head, l[s], 64);
        row[s] = has[s] ? (int64_t)((KeyT)key & row_mask) : 0;
        row_load<VEC, G, K, FULL>(w[s], tab, row[s], D, lig);
        wl[s] = tab_lin[row[s]];
        ls1[s] = ls2[s] = 0.f;
        if (OPT != OPT_SGD) {
          row_load<VEC, G, K, FULL>(s1[s], a.o.item_s1, row[s], D, lig);
          ls1[s] = a.o.item_lin_s1[row[s]];
          if (OPT == OPT_ADAM) {
            row_load<VEC, G, K, FULL>(s2[s], a.o.item_s2, row[s], D, lig);
            ls2[s] = a.o.item_lin_s2[row[s]];
          }
        }
#pragma unroll
        for (int j = 0; j < SI_MEMB; ++j) {
          const int src = l[s] + j < 64 ? l[s] + j : 63;
          const int tj = __shfl(xrow, src, 64);
          const float cj = __shfl(c, src, 64);
          cm[s][j] = (has[s] && j < len[s]) ? cj : 0.f;
          row_load<VEC, G, K, FULL>(u[s][j], a.ustage, (int64_t)tj, D, lig);
        }
      }
#pragma unroll
      for (int s = 0; s < SI_SLOTS; ++s) {
        RowReg<VEC, K> acc;
        // same order as a sequential walk of the run: c0*u0, then += c*u member by member
#pragma unroll
        for (int q = 0; q < N; ++q) acc.v[q] = cm[s][0] * u[s][0].v[q];
        float lin = cm[s][0];
#pragma unroll
        for (int j = 1; j < SI_MEMB; ++j) {
          // lanes past the run multiply by c = 0; their row bits are cleared too so that a non-finite stranger cannot leak in
          const uint32_t on = (has[s] && j < len[s]) ? 0xffffffffu : 0u;
#pragma unroll
          for (int q = 0; q < N; ++q) acc.v[q] += cm[s][j] * __uint_as_float(__float_as_uint(u[s][j].v[q]) & on);
          lin += cm[s][j];
        }
        // longer runs (hot rows, metadata columns): TU more members per turn.  The run lies inside this wave's 64
        // references, so staging rows and coefficients come from the detection lanes by shuffle — the loop is
        // wave-uniform (all lanes shuffle; groups whose run has ended add masked zeros) and its row loads are independent.
        constexpr int TU = N <= 4 ? 4 : 2;
        for (int j0 = SI_MEMB; __ballot(has[s] && j0 < len[s]) != 0ull; j0 += TU) {
          RowReg<VEC, K> uj[TU];
          float cj[TU];
          uint32_t onj[TU];
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) {
            const bool on = has[s] && j0 + tu < len[s];
            const int src = on ? l[s] + j0 + tu : l[s];
            const int tj = __shfl(xrow, src, 64);
            const float cv = __shfl(c, src, 64);
            cj[tu] = on ? cv : 0.f;
            onj[tu] = on ? 0xffffffffu : 0u;
            row_load<VEC, G, K, FULL>(uj[tu], a.ustage, (int64_t)tj, D, lig);
          }
#pragma unroll
          for (int tu = 0; tu < TU; ++tu) {
#pragma unroll
            for (int q = 0; q < N; ++q) acc.v[q] += cj[tu] * __uint_as_float(__float_as_uint(uj[tu].v[q]) & onj[tu]);
            lin += cj[tu];
          }
        }
        if (!has[s]) continue;
        const bool cut_tail = (l[s] + len[s] == RUN_CHUNK) && (base + RUN_CHUNK < n) && knext == keyv[s];
        float* irow = tab + row[s] * (int64_t)D;
        if (hd[s] != 0 && !cut_tail) {
          if (OPT == OPT_SGD) {
            const float sub = a.fmsub ? lin : 0.f;  // FM with metadata: -sum(c) * w, the "- v" of g*(S - v)
#pragma unroll
            for (int q = 0; q < N; ++q) w[s].v[q] += a.fmsub ? acc.v[q] - sub * w[s].v[q] : acc.v[q];
            row_store<VEC, G, K>(w[s], irow, D, lig);
            if (lig == 0) tab_lin[row[s]] = wl[s] + lin;
          } else {  // the run is the row's whole gradient: apply the rule once, state rows beside the weights
            const float sub = a.fmsub ? lin : 0.f;  // FM with metadata: G = sum(c*S) - sum(c)*w
#pragma unroll
            for (int q = 0; q < N; ++q)
              w[s].v[q] = opt_apply<OPT>(w[s].v[q], a.fmsub ? acc.v[q] - sub * w[s].v[q] : acc.v[q], s1[s].v[q], s2[s].v[q],
                                         a.o);
            row_store<VEC, G, K>(w[s], irow, D, lig);
            row_store<VEC, G, K>(s1[s], a.o.item_s1 + row[s] * (int64_t)D, D, lig);
            if (OPT == OPT_ADAM) row_store<VEC, G, K>(s2[s], a.o.item_s2 + row[s] * (int64_t)D, D, lig);
            if (lig == 0) {
              tab_lin[row[s]] = opt_apply<OPT>(wl[s], lin, ls1[s], ls2[s], a.o);
              a.o.item_lin_s1[row[s]] = ls1[s];
              if (OPT == OPT_ADAM) a.o.item_lin_s2[row[s]] = ls2[s];
            }
          }
        } else if (OPT != OPT_SGD) {
          // a piece of a run cut at a chunk boundary: the rule is not linear, so the pieces first meet in the (zeroed)
          // gradient accumulator; the head piece lists the row for cut_rows_apply_kernel.  (fmsub: w is the pre-update
          // row in every piece — the rule is applied after all of them — so the pieces' -sum(c)*w add up exactly)
          float* grow = a.o.gacc + row[s] * (int64_t)D;
#pragma unroll
          for (int kk = 0; kk < K; ++kk) {
            const int e = (kk * G + lig) * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q)
              if (e + q < D)
                atomicAdd(grow + e + q, a.fmsub ? acc.v[kk * VEC + q] - lin * w[s].v[kk * VEC + q] : acc.v[kk * VEC + q]);
          }
          if (lig == 0) {
            atomicAdd(a.o.gacc_lin + row[s], lin);
            if (hd[s] != 0) {
              const int slot = atomicAdd(a.o.cut_count + a.parity, 1);
              if (slot < a.o.cut_capacity) a.o.cut_rows[slot] = (int32_t)row[s];
            }
          }
        } else {  // a cut piece of a long segment (hot row): several groups add into the row
          // (fmsub: the piece subtracts sum(c)*w with the w it loaded; another piece may already have added into the
          // row, which changes this term only in second order of the step size)
#pragma unroll
          for (int kk = 0; kk < K; ++kk) {
            const int e = (kk * G + lig) * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q)
              if (e + q < D)
                atomicAdd(irow + e + q, a.fmsub ? acc.v[kk * VEC + q] - lin * w[s].v[kk * VEC + q] : acc.v[kk * VEC + q]);
          }
          if (lig == 0) atomicAdd(tab_lin + row[s], lin);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // lead_lane is rewritten by the next chunk
  }
}

// ---------------------------------------------------------------------------------------------- user duplicates
// flags[q] = 1 iff the user of position q is referenced by another triple of the same batch (static for the epoch):
// per-batch grouping of (user, q) by user, then compare neighbours inside the batch.
template <typename KeyT>
__global__ __launch_bounds__(TRS_BLOCK) void user_flags_kernel(const KeyT* __restrict__ keys,
                                                              const uint32_t* __restrict__ vals, int64_t n_pos,
                                                              int64_t batch, uint8_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t s = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; s < n_pos; s += stride) {
    const KeyT k = keys[s];
    const int64_t in_seg = s % batch;
    const bool dup = (in_seg > 0 && keys[s - 1] == k) || (in_seg + 1 < batch && keys[s + 1] == k);
    if (dup) flags[vals[s]] = 1;  // the array was zeroed: only the (few) duplicated positions take a scattered byte store
  }
}

// iflags[2q + w] = 1 iff the item row of reference w of position q has another reference in q's batch: neighbour compare
// on the batch's sorted keys, scattered back by payload (2t + w inside the batch).  The array was zeroed.
__global__ __launch_bounds__(TRS_BLOCK) void item_flags_kernel(const uint32_t* __restrict__ keys,
                                                              const RefPayload* __restrict__ vals, int64_t n_refs,
                                                              int64_t per_batch, uint8_t* __restrict__ iflags) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t s = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; s < n_refs; s += stride) {
    const uint32_t k = keys[s];
    const int64_t in_seg = s % per_batch;
    const bool dup = (in_seg > 0 && keys[s - 1] == k) || (in_seg + 1 < per_batch && keys[s + 1] == k);
    if (dup) iflags[(s - in_seg) + (int64_t)vals[s].tw] = 1;
  }
}

// K3'' : user rows referenced by several triples of the batch.  The slice's (batch, user) sort puts them next to each
// other: the first entry of a run of length > 1 sums the staged gradients of the run's triples and applies them with ONE
// plain read-modify-write (K1 already updated every user referenced once).  No atomics: the step is reproducible.
struct UserDupArgs {
  trs_tables T;
  const void* ukeys;      // this step's B sorted user ids
  const uint32_t* uvals;  // this step's B sorted positions (slice-relative q)
  int64_t B;
  int64_t q0;             // slice-relative position of the step's first triple
  int user_bits;
  const float* du;        // (B,D) staged user-row gradients (written by K1 for duplicated users)
  const float* gz;        // (2,B)
  float lr;
  OptArgs o;              // update rule (OPT_SGD: lr above)
  // ukeys == NULL (plain SGD): no sorted user runs — the flagged users (conservative flags of the LDS-bitmap kernel) add
  // their staged gradient rows into the table with float atomics (flagged_user_update_body)
  const uint8_t* uflags;    // this step's B user-duplicate flags
  const int32_t* user_ids;  // this step's B user ids
};

// Plain SGD without a user sort: one lane per position of the batch finds the flagged users; the wave then adds their
// staged gradient rows, user[u] += -lr * du[t] (+ the 1-wide term), a row = adjacent dwords per atomic instruction.  Every
// read of the step happened in K1, so this is exact up to the order of the sums on shared rows.  c2: 6 % of the users.
__device__ __forceinline__ void flagged_user_update_body(const UserDupArgs& a, int block_id, int n_blocks) {
  const trs_tables& T = a.T;
  const int D = T.D;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)block_id * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)n_blocks * TRS_BLOCK) >> 6;
  for (int64_t base = wave * TRS_WAVE; base < a.B; base += nwave * TRS_WAVE) {
    const int64_t t = base + lane;
    const bool valid = t < a.B;
    const int64_t tc = valid ? t : a.B - 1;
    const int32_t u = a.user_ids[tc];
    const float clin = -a.lr * (a.gz[tc] + a.gz[a.B + tc]);
    uint64_t mask = __ballot(valid && a.uflags[tc] != 0);
    while (mask) {  // four flagged users per turn: their staged rows are loaded together (small batches over small
      // tables flag a third of the positions — c1: one row per turn cost 6 us per step)
      int l[4];
      bool on[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        on[k] = mask != 0;
        l[k] = on[k] ? __ffsll((unsigned long long)mask) - 1 : 0;
        mask &= mask - 1;  // (0 stays 0)
      }
      float v[4];
      int64_t uk[4];
      float cl[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uk[k] = __shfl(u, l[k], 64);
        cl[k] = __shfl(clin, l[k], 64);
        v[k] = a.du[(base + l[k]) * (int64_t)D + (lane < D ? lane : 0)];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!on[k]) continue;
        float* dst = T.user + uk[k] * (int64_t)D;
        if (lane < D) atomicAdd(dst + lane, -a.lr * v[k]);
        for (int e = lane + TRS_WAVE; e < D; e += TRS_WAVE) atomicAdd(dst + e, -a.lr * a.du[(base + l[k]) * (int64_t)D + e]);
        if (lane == 0) atomicAdd(T.user_lin + uk[k], cl[k]);
      }
    }
  }
}

template <typename KeyT, int VEC, int G, int K, bool FULL, int OPT = OPT_SGD>
__device__ __forceinline__ void sorted_user_dup_update_body(const UserDupArgs& a, int block_id, int n_blocks) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  const trs_tables& T = a.T;
  const int D = T.D;
  const int64_t n = a.B;
  const KeyT* keys = reinterpret_cast<const KeyT*>(a.ukeys);
  const KeyT user_mask = (KeyT)(((uint64_t)1 << a.user_bits) - 1);
  const int lane = threadIdx.x & 63;
  const int lig = lane % G;
  const int64_t wave = ((int64_t)block_id * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)n_blocks * TRS_BLOCK) >> 6;
  const int gi = lane / G;
  // detection: one lane per sorted entry (64 entries per wave), keys and positions in one round of loads; the few run
  // leaders are then spread over the wave's TPW lane groups, which fetch the run's staged gradients and the user row
  // in a second round (positions come from the detection lanes by shuffle, so nothing is loaded under a branch).
  const int64_t niter = (n + TRS_WAVE - 1) / TRS_WAVE;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t i = it * TRS_WAVE + lane;
    const bool valid = i < n;
    const int64_t il = valid ? i : n - 1;
    const KeyT k0 = keys[il];
    const KeyT kp = keys[il > 0 ? il - 1 : 0];
    const KeyT kn = keys[il + 1 < n ? il + 1 : il];
    const int uv = (int)((int64_t)a.uvals[il] - a.q0);
    const bool cont = valid && il > 0 && kp == k0;
    const bool lead = valid && !cont && (il + 1 < n && kn == k0);
    const uint64_t lmask = __ballot(lead);
    const uint64_t cmask = __ballot(cont);
    const int nlead = __popcll(lmask);
    for (int r = 0; r * TPW < nlead; ++r) {
      const int want = r * TPW + gi;
      uint64_t m = lmask;
      for (int q = 0; q < want && m; ++q) m &= m - 1;
      const bool has = m != 0;
      const int l = has ? __ffsll((unsigned long long)m) - 1 : 0;
      // members of the run inside this wave: lane l and the cont lanes right after it
      const uint64_t after = l < 63 ? (cmask >> (l + 1)) : 0ull;
      const int in_wave = has ? 1 + (int)__ffsll((unsigned long long)~after) - 1 : 0;  // ~after != 0: shifted in zeros
      int most = in_wave;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(most, o, 64);
        most = other > most ? other : most;
      }
      const int64_t key_lo = __shfl((int)(uint32_t)(k0 & (KeyT)0xffffffffu), l, 64);
      const int64_t user = (int64_t)((uint32_t)key_lo) & (int64_t)user_mask;
      RowReg<VEC, K> w, acc, s1, s2;
      row_load<VEC, G, K, FULL>(w, T.user, user, D, lig);
      const float wl = T.user_lin[user];
      float ls1 = 0.f, ls2 = 0.f;
      if (OPT != OPT_SGD) {
        row_load<VEC, G, K, FULL>(s1, a.o.user_s1, user, D, lig);
        ls1 = a.o.user_lin_s1[user];
        if (OPT == OPT_ADAM) {
          row_load<VEC, G, K, FULL>(s2, a.o.user_s2, user, D, lig);
          ls2 = a.o.user_lin_s2[user];
        }
      }
#pragma unroll
      for (int q = 0; q < N; ++q) acc.v[q] = 0.f;
      float lin = 0.f;
      for (int j = 0; j < most; ++j) {
        const int src = l + j < 64 ? l + j : 63;
        const int t = __shfl(uv, src, 64);
        // bit mask, not a multiply: rows of du outside duplicate runs are never written and may hold anything
        const uint32_t on = (has && j < in_wave) ? 0xffffffffu : 0u;
        RowReg<VEC, K> g;
        row_load<VEC, G, K, FULL>(g, a.du, (int64_t)t, D, lig);
#pragma unroll
        for (int q = 0; q < N; ++q) acc.v[q] += __uint_as_float(__float_as_uint(g.v[q]) & on);
        lin += __uint_as_float(__float_as_uint(a.gz[t] + a.gz[a.B + t]) & on);
      }
      if (!has) continue;
      if (l + in_wave == TRS_WAVE) {  // rare: the run goes on in the next 64 entries
        const KeyT key = keys[it * TRS_WAVE + l];
        for (int64_t j = (it + 1) * TRS_WAVE; j < n && keys[j] == key; ++j) {
          const int64_t t = (int64_t)a.uvals[j] - a.q0;
          RowReg<VEC, K> g;
          row_load<VEC, G, K, FULL>(g, a.du, t, D, lig);
#pragma unroll
          for (int q = 0; q < N; ++q) acc.v[q] += g.v[q];
          lin += a.gz[t] + a.gz[a.B + t];
        }
      }
      if (OPT == OPT_SGD) {
#pragma unroll
        for (int q = 0; q < N; ++q) w.v[q] += (-a.lr) * acc.v[q];
        row_store<VEC, G, K>(w, T.user + user * (int64_t)D, D, lig);
        if (lig == 0) T.user_lin[user] = wl + (-a.lr) * lin;
      } else {
#pragma unroll
        for (int q = 0; q < N; ++q) w.v[q] = opt_apply<OPT>(w.v[q], acc.v[q], s1.v[q], s2.v[q], a.o);
        row_store<VEC, G, K>(w, T.user + user * (int64_t)D, D, lig);
        row_store<VEC, G, K>(s1, a.o.user_s1 + user * (int64_t)D, D, lig);
        if (OPT == OPT_ADAM) row_store<VEC, G, K>(s2, a.o.user_s2 + user * (int64_t)D, D, lig);
        if (lig == 0) {
          T.user_lin[user] = opt_apply<OPT>(wl, lin, ls1, ls2, a.o);
          a.o.user_lin_s1[user] = ls1;
          if (OPT == OPT_ADAM) a.o.user_lin_s2[user] = ls2;
        }
      }
    }
  }
}

// Item rows whose run was cut at a chunk boundary (adaptive rules): the pieces' sums are in gacc; apply the rule once
// per listed row, clear the accumulator row, and reset the OTHER step parity's counter for the next step.
template <int VEC, int G, int K, bool FULL, int OPT>
__global__ __launch_bounds__(TRS_BLOCK) void cut_rows_apply_kernel(const trs_tables T, const OptArgs o, int parity,
                                                                  float* tab, float* tab_lin) {
  constexpr int N = K * VEC;
  constexpr int TPW = TRS_WAVE / G;
  const int D = T.D;
  if (!tab) {  // the item table, or one metadata column's tables (o.item_* are then that column's state)
    tab = T.item;
    tab_lin = T.item_lin;
  }
  const int lane = threadIdx.x & 63, lig = lane % G;
  int count = o.cut_count[parity];
  if (count > o.cut_capacity) count = o.cut_capacity;
  if (blockIdx.x == 0 && threadIdx.x == 0) o.cut_count[parity ^ 1] = 0;
  const int64_t wave = ((int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * TRS_BLOCK) >> 6;
  for (int64_t i = wave * TPW + lane / G; i < count; i += nwave * TPW) {
    const int64_t row = o.cut_rows[i];
    RowReg<VEC, K> w, g, s1, s2, z;
    row_load<VEC, G, K, FULL>(w, tab, row, D, lig);
    row_load<VEC, G, K, FULL>(g, o.gacc, row, D, lig);
    row_load<VEC, G, K, FULL>(s1, o.item_s1, row, D, lig);
    if (OPT == OPT_ADAM) row_load<VEC, G, K, FULL>(s2, o.item_s2, row, D, lig);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      w.v[q] = opt_apply<OPT>(w.v[q], g.v[q], s1.v[q], s2.v[q], o);
      z.v[q] = 0.f;
    }
    row_store<VEC, G, K>(w, tab + row * (int64_t)D, D, lig);
    row_store<VEC, G, K>(s1, o.item_s1 + row * (int64_t)D, D, lig);
    if (OPT == OPT_ADAM) row_store<VEC, G, K>(s2, o.item_s2 + row * (int64_t)D, D, lig);
    row_store<VEC, G, K>(z, o.gacc + row * (int64_t)D, D, lig);
    if (lig == 0) {
      float ls1 = o.item_lin_s1[row], ls2 = OPT == OPT_ADAM ? o.item_lin_s2[row] : 0.f;
      tab_lin[row] = opt_apply<OPT>(tab_lin[row], o.gacc_lin[row], ls1, ls2, o);
      o.item_lin_s1[row] = ls1;
      if (OPT == OPT_ADAM) o.item_lin_s2[row] = ls2;
      o.gacc_lin[row] = 0.f;
    }
  }
}

template <typename KeyT, int VEC, int G, int K, bool FULL, int OPT = OPT_SGD>
__global__ __launch_bounds__(TRS_BLOCK)
__attribute__((amdgpu_waves_per_eu(OPT == OPT_SGD && VEC == 4 && K == 1 && FULL && sizeof(KeyT) == 4 ? K2_WAVES : 1)))
void sorted_item_update_staged_kernel(const SortedArgs a) {
  sorted_item_update_staged_body<KeyT, VEC, G, K, FULL, OPT>(a, blockIdx.x, gridDim.x);
}

template <typename KeyT, int VEC, int G, int K, bool FULL>
__global__ __launch_bounds__(TRS_BLOCK) void sorted_user_dup_update_kernel(const UserDupArgs a) {
  sorted_user_dup_update_body<KeyT, VEC, G, K, FULL>(a, blockIdx.x, gridDim.x);
}

// Both updates of a presorted step in ONE launch (they touch different tables and only depend on K1): the first
// n_user_blocks workgroups walk the duplicated-user runs, the rest the item chunks.  Saves a kernel boundary and hides
// the short, latency-bound user pass under the item pass.  32-bit keys on both sides (the common case).
template <int VEC, int G, int K, bool FULL, int OPT = OPT_SGD>
__global__ __launch_bounds__(TRS_BLOCK)
__attribute__((amdgpu_waves_per_eu(OPT == OPT_SGD && VEC == 4 && K == 1 && FULL ? K2_WAVES : 1)))
void sorted_updates_fused_kernel(const SortedArgs ia, const UserDupArgs ua,
                                                                        int n_user_blocks) {
  if ((int)blockIdx.x < n_user_blocks) {
    if (OPT == OPT_SGD && ua.ukeys == nullptr) flagged_user_update_body(ua, blockIdx.x, n_user_blocks);
    else sorted_user_dup_update_body<uint32_t, VEC, G, K, FULL, OPT>(ua, blockIdx.x, n_user_blocks);
  } else
    sorted_item_update_staged_body<uint32_t, VEC, G, K, FULL, OPT>(ia, (int)blockIdx.x - n_user_blocks,
                                                                   (int)gridDim.x - n_user_blocks);
}

// Grid of epoch_refs_kernel.  It runs on a side stream BESIDE the training steps and is a chain of random 8-byte reads:
// launched as wide as the chip (4096 workgroups of 256 threads, 31 VGPRs: 8 per CU = every wave slot) it finishes a
// 512-batch slice in 0.4 ms — and for those 0.4 ms no workgroup of a step kernel finds a wave slot (rocprofv3, c4: step
// launches of 280 us beside a 33 us median, every slice).  Two workgroups per CU leave three quarters of the slots to
// the steps; the slice's ids then take ~4x as long, of the 18 ms the slice's steps run.
static inline int presort_grid(int64_t n_pos) {
  // (short slices — up to 2^20 positions — take a thread per triple: the launch is over in ~20 us either way, and with
  // four dependent stream reads per thread it lasted 54 us beside the steps of a 16-batch slice)
  const int64_t cap = n_pos <= ((int64_t)1 << 20) ? 4096 : trs_tuning().presort_grid_cap;
  int64_t g = (n_pos + TRS_BLOCK - 1) / TRS_BLOCK;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static int bits_for(int64_t n) {
  int b = 1;
  while (b < 63 && ((int64_t)1 << b) < n) ++b;
  return b;
}

// batch_group_items_kernel<SRC> for n_batches batches of `batch` positions over a table of n_rows rows.  pay_bits = bits
// of the largest payload (SRC 0 / 1: 2 * batch references, SRC 2: batch).  0, or a negative error.
template <int SRC>
static int launch_group(const int32_t* pos, const int32_t* neg, int64_t n_batches, int64_t batch, int64_t n_rows,
                        int pay_bits, uint32_t* keys_out, RefPayload* vals_out, hipStream_t s, const char* who) {
  const int refs_bits = bits_for((SRC == 2 ? 1 : 2) * batch);
  int bin_bits = 32 - pay_bits;
  if (bin_bits > GRP_MAX_BIN_BITS) bin_bits = GRP_MAX_BIN_BITS;
  if (bin_bits < GRP_MIN_BIN_BITS || ((int64_t)1 << refs_bits) / GRP_CAP + 2 > GRP_WMAX) {
    trs_set_error("%s: a batch of %lld positions is too long for the per-batch grouping (at most 2^21 references)", who,
                  (long long)batch);
    return TRS_E_ARG;
  }
  const size_t lds = (size_t)GRP_LDS_WORDS * 4;
  const dim3 gr((unsigned)n_batches), bl(GRP_THREADS);
#define TRS_GRP(BB)                                                                                                 \
  case BB: {                                                                                                        \
    static const int attr_done = [] { /* > 64 KB of dynamic LDS needs the opt-in, once per kernel (thread-safe static) */ \
      (void)hipFuncSetAttribute((const void*)batch_group_items_kernel<SRC, BB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                GRP_LDS_WORDS * 4);                                                                  \
      return 1;                                                                                                     \
    }();                                                                                                            \
    (void)attr_done;                                                                                                \
    hipLaunchKernelGGL((batch_group_items_kernel<SRC, BB>), gr, bl, lds, s, pos, neg, batch, n_rows, pay_bits, keys_out, \
                       vals_out);                                                                                   \
    break;                                                                                                          \
  }
  switch (bin_bits) {
    TRS_GRP(14)
    TRS_GRP(13)
    TRS_GRP(12)
    TRS_GRP(11)
    TRS_GRP(10)
  }
#undef TRS_GRP
  TRS_CHECK_LAUNCH("batch_group_items_kernel");
  return TRS_OK;
}

}  // namespace trs

using namespace trs;

// Sizes (bytes) a caller must provide for an epoch slice of n_batches whole batches:
//   ids: 3 x n_pos int32 (user, pos, neg)   keys: 2 x (2 n_pos) keys   vals: 2 x (2 n_pos) x 8   temp: 256 bytes, unused
extern "C" int trs_epoch_presort_sizes(int64_t n_batches, int64_t batch, int64_t n_items, int64_t* key_bytes_out,
                                       int64_t* keys_total_bytes_out, int64_t* vals_total_bytes_out,
                                       int64_t* temp_bytes_out) {
  TRS_REQUIRE(n_batches > 0 && batch > 0 && n_items > 0, "trs_epoch_presort_sizes: bad arguments");
  TRS_REQUIRE(key_bytes_out && keys_total_bytes_out && vals_total_bytes_out && temp_bytes_out,
              "trs_epoch_presort_sizes: NULL output");
  const int bits = bits_for(n_items);
  TRS_REQUIRE(bits <= 32 && 2 * n_batches * batch < ((int64_t)1 << 32), "trs_epoch_presort_sizes: slice too long");
  const int kb = 4;
  const size_t n = (size_t)(2 * n_batches * batch);
  const size_t temp = 0;  // (no scratch since the vendor sort is gone; the argument stays for the ABI)
  *key_bytes_out = kb;
  *keys_total_bytes_out = 2 * (int64_t)n * kb;
  *vals_total_bytes_out = 2 * (int64_t)n * (int64_t)sizeof(RefPayload);
  *temp_bytes_out = (int64_t)temp + 256;
  return TRS_OK;
}

// Builds the epoch slice: ids (user/pos/neg, n_pos each; generated when stream_ui is non-NULL, else taken as given and
// only clamped) and the item references sorted per batch by item row.  keys_dev / vals_dev hold two halves each (input |
// sorted output); *sorted_keys_out / *sorted_vals_out receive the device addresses of the sorted halves.
extern "C" int trs_epoch_presort(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N,
                                 uint64_t shuffle_key, uint64_t sample_seed, int64_t first_pos, int64_t n_batches,
                                 int64_t batch, int64_t n_users, int64_t n_items, int32_t* user_dev, int32_t* pos_dev,
                                 int32_t* neg_dev, void* keys_dev, void* vals_dev, void* temp_dev, int64_t temp_bytes,
                                 int32_t* err_flag_dev, void** sorted_keys_out, void** sorted_vals_out,
                                 uint8_t* item_dup_flags_out_dev, const trs_sampler* sampler, void* stream) {
  TRS_REQUIRE(n_batches > 0 && batch > 0 && n_users > 0 && n_items > 0, "trs_epoch_presort: bad sizes");
  TRS_REQUIRE(!sampler || (sampler->k_neg >= 1 && (!sampler->popularity || (sampler->pop_items && sampler->pop_n > 0)) &&
                           ((sampler->seen_off == nullptr) == (sampler->seen_items == nullptr))),
              "trs_epoch_presort: bad sampler options");
  TRS_REQUIRE(user_dev && pos_dev && neg_dev && keys_dev && vals_dev && temp_dev, "trs_epoch_presort: NULL buffer");
  TRS_REQUIRE(sorted_keys_out && sorted_vals_out, "trs_epoch_presort: NULL output");
  const int64_t n_pos = n_batches * batch;
  const int64_t kn = sampler && sampler->k_neg > 1 ? sampler->k_neg : 1;
  if (stream_ui_dev)
    TRS_REQUIRE(N > 0 && first_pos >= 0 && first_pos + n_pos <= N * kn, "trs_epoch_presort: slice outside the stream");
  EpochArgs a = {};
  a.S = trs_sampler_args(sampler);
  a.sui = (const int2*)stream_ui_dev;
  a.neg_static = neg_static_dev;
  a.N = N;
  a.shuffle_key = shuffle_key;
  a.hb = stream_ui_dev ? trs_feistel_half_bits(N * kn) : 0;
  a.sample_seed = sample_seed;
  a.first_pos = first_pos;
  a.n_pos = n_pos;
  a.batch = batch;
  a.n_users = n_users;
  a.n_items = n_items;
  a.item_bits = bits_for(n_items);
  a.user = user_dev;
  a.pos = pos_dev;
  a.neg = neg_dev;
  a.keys = keys_dev;
  a.vals = (RefPayload*)vals_dev;
  a.err = err_flag_dev;
  const int bits = a.item_bits;
  TRS_REQUIRE(bits <= 32 && 2 * n_pos < ((int64_t)1 << 32), "trs_epoch_presort: slice too long");
  const int src = !stream_ui_dev ? 0 : (neg_static_dev ? 2 : 1);
  hipStream_t s = (hipStream_t)stream;
  const dim3 gr(presort_grid(n_pos)), bl(TRS_BLOCK);
  const int pay_bits = bits_for(2 * batch);
  a.keys = nullptr;  // the grouping kernel reads pos / neg themselves
  if (src == 0) hipLaunchKernelGGL((epoch_refs_kernel<uint32_t, 0>), gr, bl, 0, s, a);
  else if (src == 1) hipLaunchKernelGGL((epoch_refs_kernel<uint32_t, 1>), gr, bl, 0, s, a);
  else hipLaunchKernelGGL((epoch_refs_kernel<uint32_t, 2>), gr, bl, 0, s, a);
  TRS_CHECK_LAUNCH("epoch_refs_kernel");
  const size_t n = (size_t)(2 * n_pos);
  (void)temp_bytes;
  RefPayload* vin = (RefPayload*)vals_dev;
  RefPayload* vout = vin + n;
  uint32_t* kin = (uint32_t*)keys_dev;
  // hand-written grouping (counting sort in LDS, one workgroup per batch): no key array, no vendor library
  const int rc = launch_group<0>((const int32_t*)pos_dev, (const int32_t*)neg_dev, n_batches, batch, n_items, pay_bits,
                                 kin + n, vout, s, "trs_epoch_presort");
  if (rc) return rc;
  *sorted_keys_out = (void*)(kin + n);
  *sorted_vals_out = (void*)vout;
  if (item_dup_flags_out_dev) {
    (void)hipMemsetAsync(item_dup_flags_out_dev, 0, n, s);
    hipLaunchKernelGGL(item_flags_kernel, dim3(trs_grid((int64_t)n, TRS_BLOCK)), bl, 0, s, kin + n, vout, (int64_t)n,
                       2 * batch, item_dup_flags_out_dev);
    TRS_CHECK_LAUNCH("item_flags_kernel");
  }
  return TRS_OK;
}

// batch_flags_kernel<NG> with as many 64-bit words of "came later" bits per thread as the batch needs
static int launch_flags(const FlagArgs& a, int64_t n_batches, size_t lds, hipStream_t s) {
  const int64_t per_group = (int64_t)FLAG_THREADS * FLAG_U * FLAG_ROUNDS;
  const int ng = (int)((a.batch + per_group - 1) / per_group);
  const dim3 gr((unsigned)n_batches), bl(FLAG_THREADS);
#define TRS_FL(NG)                                                                                                   \
  {                                                                                                                  \
    static const int attr_done = [] { /* > 64 KB of dynamic LDS needs the opt-in once per kernel */                   \
      (void)hipFuncSetAttribute((const void*)batch_flags_kernel<NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
      return 1;                                                                                                      \
    }();                                                                                                             \
    (void)attr_done;                                                                                                 \
    hipLaunchKernelGGL((batch_flags_kernel<NG>), gr, bl, lds, s, a);                                                  \
  }
  if (ng <= 1) TRS_FL(1) else if (ng == 2) TRS_FL(2) else TRS_FL(4)
#undef TRS_FL
  TRS_CHECK_LAUNCH("batch_flags_kernel");
  return TRS_OK;
}

// Ids + conservative duplicate flags of an epoch slice in one launch (the sparse regime's whole presort).
extern "C" int trs_epoch_flags_ordered(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N,
                                       uint64_t shuffle_key, uint64_t sample_seed, int64_t first_pos, int64_t n_batches,
                                       int64_t batch, int64_t n_users, int64_t n_items, int32_t* user_dev,
                                       int32_t* pos_dev, int32_t* neg_dev, uint8_t* user_dup_flags_out_dev,
                                       uint8_t* item_dup_flags_out_dev, int32_t* n_flagged_out_dev,
                                       int32_t* err_flag_dev, const trs_sampler* sampler, void* stream);

extern "C" int trs_epoch_flags(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N,
                               uint64_t shuffle_key, uint64_t sample_seed, int64_t first_pos, int64_t n_batches,
                               int64_t batch, int64_t n_users, int64_t n_items, int32_t* user_dev, int32_t* pos_dev,
                               int32_t* neg_dev, uint8_t* user_dup_flags_out_dev, uint8_t* item_dup_flags_out_dev,
                               int32_t* err_flag_dev, const trs_sampler* sampler, void* stream) {
  return trs_epoch_flags_ordered(stream_ui_dev, neg_static_dev, N, shuffle_key, sample_seed, first_pos, n_batches, batch,
                                 n_users, n_items, user_dev, pos_dev, neg_dev, user_dup_flags_out_dev,
                                 item_dup_flags_out_dev, nullptr, err_flag_dev, sampler, stream);
}

// trs_epoch_flags + (n_flagged_out_dev != NULL) the flagged-first order of every batch: see batch_flags_kernel, phase D.
extern "C" int trs_epoch_flags_ordered(const int32_t* stream_ui_dev, const int32_t* neg_static_dev, int64_t N,
                                       uint64_t shuffle_key, uint64_t sample_seed, int64_t first_pos, int64_t n_batches,
                                       int64_t batch, int64_t n_users, int64_t n_items, int32_t* user_dev,
                                       int32_t* pos_dev, int32_t* neg_dev, uint8_t* user_dup_flags_out_dev,
                                       uint8_t* item_dup_flags_out_dev, int32_t* n_flagged_out_dev,
                                       int32_t* err_flag_dev, const trs_sampler* sampler, void* stream) {
  const int64_t kn = sampler && sampler->k_neg > 1 ? sampler->k_neg : 1;
  TRS_REQUIRE(!sampler || (sampler->k_neg >= 1 && (!sampler->popularity || (sampler->pop_items && sampler->pop_n > 0)) &&
                           ((sampler->seen_off == nullptr) == (sampler->seen_items == nullptr))),
              "trs_epoch_flags: bad sampler options");
  TRS_REQUIRE(batch <= (int64_t)FLAG_THREADS * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS,
              "trs_epoch_flags: batch %lld exceeds %d (one workgroup per batch keeps a bit per reference in registers)",
              (long long)batch, FLAG_THREADS * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS);
  TRS_REQUIRE(n_batches > 0 && batch > 0 && batch < ((int64_t)1 << 30) && n_users > 0 && n_items > 0 &&
                  n_users < ((int64_t)1 << 31) && n_items < ((int64_t)1 << 31), "trs_epoch_flags: bad sizes");
  TRS_REQUIRE(user_dev && pos_dev && neg_dev && user_dup_flags_out_dev && item_dup_flags_out_dev,
              "trs_epoch_flags: NULL buffer");
  if (stream_ui_dev)
    TRS_REQUIRE(N > 0 && first_pos >= 0 && first_pos + n_batches * batch <= N * kn,
                "trs_epoch_flags: slice outside the stream");
  FlagArgs a = {};
  a.sui = (const int2*)stream_ui_dev;
  a.neg_static = neg_static_dev;
  a.N = N;
  a.shuffle_key = shuffle_key;
  a.hb = stream_ui_dev ? trs_feistel_half_bits(N) : 0;
  a.sample_seed = sample_seed;
  a.first_pos = first_pos;
  a.batch = batch;
  a.n_users = n_users;
  a.n_items = n_items;
  a.user = user_dev;
  a.pos = pos_dev;
  a.neg = neg_dev;
  a.uflags = user_dup_flags_out_dev;
  a.iflags = item_dup_flags_out_dev;
  a.err = err_flag_dev;
  // bitmap: as many bits as the larger table needs, between 2^10 and 2^20 (128 KB of LDS)
  const int64_t big = n_users > n_items ? n_users : n_items;
  int lb = 10;
  while (lb < 20 && ((int64_t)1 << lb) < big) ++lb;
  a.log2_bits = lb;
  a.item_hash = n_items > ((int64_t)1 << lb) ? 1 : 0;
  a.user_hash = n_users > ((int64_t)1 << lb) ? 1 : 0;
  a.what_first = 0;
  a.what_end = 2;
  size_t lds = ((size_t)1 << lb) / 8;
  if (n_flagged_out_dev) {  // phase D's scratch shares the bitmap's LDS: cell prefixes + two lists of positions
    a.nflag = n_flagged_out_dev;
    a.ord_off = (int)(((batch + 63) / 64 + 64) & ~(int64_t)63);
    const int64_t half = (batch + 1) / 2;
    a.ord_cap = (int)(half < 12288 ? (half < 64 ? 64 : half) : 12288);
    const size_t need = ((size_t)a.ord_off + 2 * (size_t)a.ord_cap) * 4;
    if (need > lds) lds = need;
  }
  int src = !stream_ui_dev ? 0 : (neg_static_dev ? 2 : 1);
  hipStream_t s = (hipStream_t)stream;
  if (src != 0) {
    // The ids first, by a WIDE launch (a thread per triple: the shuffle's 8-byte reads out of the resident stream cost a
    // 64-byte sector each and want every CU's loads in flight — 0.43 ms per 16.7M triples); the per-batch bitmap
    // kernel then reads them back contiguously.  One workgroup per batch doing both took 216 us per batch whatever
    // the slice length; split, a 16-batch slice takes ~60 us.
    EpochArgs e = {};
    e.sui = a.sui; e.neg_static = a.neg_static; e.N = a.N; e.shuffle_key = a.shuffle_key; e.hb = a.hb;
    e.sample_seed = a.sample_seed; e.first_pos = a.first_pos; e.n_pos = n_batches * batch; e.batch = batch;
    e.n_users = n_users; e.n_items = n_items; e.user = user_dev; e.pos = pos_dev; e.neg = neg_dev; e.keys = nullptr;
    e.err = err_flag_dev;
    e.S = trs_sampler_args(sampler);
    e.hb = trs_feistel_half_bits(N * kn);
    const dim3 ge(presort_grid(e.n_pos)), be(TRS_BLOCK);
    if (src == 1) hipLaunchKernelGGL((epoch_refs_kernel<uint32_t, 1>), ge, be, 0, s, e);
    else hipLaunchKernelGGL((epoch_refs_kernel<uint32_t, 2>), ge, be, 0, s, e);
    TRS_CHECK_LAUNCH("epoch_refs_kernel");
    src = 0;  // the ids now exist (validated and clamped): the flags kernel takes them as given
  }
  (void)src;  // the ids exist by now (given, or written by epoch_refs_kernel above): the flags kernel only reads them
  const int rcf = launch_flags(a, n_batches, lds, s);
  if (rcf) return rcf;
  TRS_CHECK_LAUNCH("batch_flags_kernel");
  return TRS_OK;
}

// User-duplicate flags alone (conservative when n_users exceeds the bitmap): the dense regime's plain-SGD step needs no
// sorted user runs (flagged users add their staged gradient with float atomics), so the per-batch grouping of the users of the
// slice is replaced by the bitmap kernel restricted to the user ids.
extern "C" int trs_epoch_user_flags(const int32_t* user_dev, int64_t n_batches, int64_t batch, int64_t n_users,
                                    uint8_t* flags_out_dev, void* stream) {
  TRS_REQUIRE(user_dev && flags_out_dev && n_batches > 0 && batch > 0 && n_users > 0 && n_users < ((int64_t)1 << 31),
              "trs_epoch_user_flags: bad arguments");
  TRS_REQUIRE(batch <= (int64_t)FLAG_THREADS * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS,
              "trs_epoch_user_flags: batch %lld exceeds %d", (long long)batch,
              FLAG_THREADS * FLAG_U * FLAG_ROUNDS * FLAG_MAX_GROUPS);
  FlagArgs a = {};
  a.batch = batch;
  a.n_users = n_users;
  a.n_items = n_users;  // (pos / neg alias the user ids below: they must pass the id check, nothing is written back)
  a.user = const_cast<int32_t*>(user_dev);  // SRC 0 with valid ids: read only
  a.pos = a.neg = a.user;
  a.uflags = flags_out_dev;
  int lb = 10;
  while (lb < 20 && ((int64_t)1 << lb) < n_users) ++lb;
  a.log2_bits = lb;
  a.user_hash = n_users > ((int64_t)1 << lb) ? 1 : 0;
  a.what_first = 1;
  a.what_end = 2;
  const int rcf = launch_flags(a, n_batches, ((size_t)1 << lb) / 8, (hipStream_t)stream);
  if (rcf) return rcf;
  TRS_CHECK_LAUNCH("batch_flags_kernel");
  return TRS_OK;
}

// User-duplicate flags of an epoch slice (n_pos = n_batches*batch positions; position q < 2^32).  ukeys: 2 x n_pos keys
// (4 B), uvals: 2 x n_pos uint32, temp: 256 bytes, unused.
extern "C" int trs_epoch_user_dups_sizes(int64_t n_batches, int64_t batch, int64_t n_users, int64_t* ukeys_bytes_out,
                                         int64_t* uvals_bytes_out, int64_t* temp_bytes_out) {
  TRS_REQUIRE(n_batches > 0 && batch > 0 && n_users > 0 && ukeys_bytes_out && uvals_bytes_out && temp_bytes_out,
              "trs_epoch_user_dups_sizes: bad arguments");
  TRS_REQUIRE(n_batches * batch < ((int64_t)1 << 32), "trs_epoch_user_dups_sizes: slice too long");
  const int bits = bits_for(n_users);
  TRS_REQUIRE(bits <= 32, "trs_epoch_user_dups_sizes: user ids do not fit 32 bits");
  const int kb = 4;
  const size_t n = (size_t)(n_batches * batch);
  const size_t temp = 0;  // (no scratch: the argument stays for the ABI)
  *ukeys_bytes_out = 2 * (int64_t)n * kb;
  *uvals_bytes_out = 2 * (int64_t)n * 4;
  *temp_bytes_out = (int64_t)temp + 256;
  return TRS_OK;
}

extern "C" int trs_epoch_user_dups(const int32_t* user_dev, int64_t n_batches, int64_t batch, int64_t n_users,
                                   void* ukeys_dev, void* uvals_dev, void* temp_dev, int64_t temp_bytes,
                                   uint8_t* flags_out_dev, void** sorted_ukeys_out, void** sorted_uvals_out,
                                   int32_t* ukey_bytes_out, void* stream) {
  TRS_REQUIRE(user_dev && ukeys_dev && uvals_dev && temp_dev && flags_out_dev, "trs_epoch_user_dups: NULL buffer");
  TRS_REQUIRE(n_batches > 0 && batch > 0 && n_users > 0 && n_batches * batch < ((int64_t)1 << 32),
              "trs_epoch_user_dups: bad sizes");
  const int64_t n_pos = n_batches * batch;
  const int user_bits = bits_for(n_users);
  TRS_REQUIRE(user_bits <= 32, "trs_epoch_user_dups: user ids do not fit 32 bits");
  hipStream_t s = (hipStream_t)stream;
  const dim3 gr(trs_grid(n_pos, TRS_BLOCK)), bl(TRS_BLOCK);
  (void)temp_bytes;
  uint32_t* vin = (uint32_t*)uvals_dev;
  uint32_t* kin = (uint32_t*)ukeys_dev;
  // one batch per workgroup: the batch's users grouped by the counting sort (one reference per position; the user ids are
  // read where they lie, the payload is the position in the slice)
  const int rc = launch_group<2>(user_dev, (const int32_t*)nullptr, n_batches, batch, n_users, bits_for(batch),
                                 kin + n_pos, (RefPayload*)(vin + n_pos), s, "trs_epoch_user_dups");
  if (rc) return rc;
  (void)hipMemsetAsync(flags_out_dev, 0, (size_t)n_pos, s);
  hipLaunchKernelGGL((user_flags_kernel<uint32_t>), gr, bl, 0, s, kin + n_pos, vin + n_pos, n_pos, batch, flags_out_dev);
  if (sorted_ukeys_out) *sorted_ukeys_out = (void*)(kin + n_pos);
  if (ukey_bytes_out) *ukey_bytes_out = 4;
  if (sorted_uvals_out) *sorted_uvals_out = (void*)(vin + n_pos);
  TRS_CHECK_LAUNCH("user_flags_kernel");
  return TRS_OK;
}

// References of one metadata column: key = metadata id of the reference's item, payload as for the item references.
__global__ __launch_bounds__(TRS_BLOCK) void meta_refs_kernel(const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ neg, int64_t n_pos,
                                                             int64_t batch, const int32_t* __restrict__ item_meta, int M,
                                                             int m, int64_t n_cat, uint32_t* __restrict__ keys,
                                                             RefPayload* __restrict__ vals, int32_t* err,
                                                             int32_t* __restrict__ pos_meta_out,
                                                             int32_t* __restrict__ neg_meta_out) {
  const int64_t stride = (int64_t)gridDim.x * TRS_BLOCK;
  for (int64_t q = (int64_t)blockIdx.x * TRS_BLOCK + threadIdx.x; q < n_pos; q += stride) {
    int64_t kp = item_meta[(int64_t)pos[q] * M + m], kn = item_meta[(int64_t)neg[q] * M + m];
    if ((uint64_t)kp >= (uint64_t)n_cat || (uint64_t)kn >= (uint64_t)n_cat) {
      if (err) atomicOr(err, 1);
      if ((uint64_t)kp >= (uint64_t)n_cat) kp = 0;
      if ((uint64_t)kn >= (uint64_t)n_cat) kn = 0;
    }
    keys[2 * q] = (uint32_t)kp;
    keys[2 * q + 1] = (uint32_t)kn;
    if (pos_meta_out) {  // (n_pos, M) id arrays for K1: contiguous reads instead of a lookup behind the item id
      pos_meta_out[q * M + m] = (int32_t)kp;
      neg_meta_out[q * M + m] = (int32_t)kn;
    }
  }
}

// Sorted references of metadata column m for an epoch slice whose ids (pos, neg: already validated item ids) exist:
// same buffers / sizes as trs_epoch_presort with n_items := n_cat.
extern "C" int trs_epoch_presort_meta(const int32_t* pos_dev, const int32_t* neg_dev, int64_t n_batches, int64_t batch,
                                      const int32_t* item_meta_dev, int32_t M, int32_t m, int64_t n_cat,
                                      void* keys_dev, void* vals_dev, void* temp_dev, int64_t temp_bytes,
                                      int32_t* err_flag_dev, void** sorted_keys_out, void** sorted_vals_out,
                                      int32_t* pos_meta_out_dev, int32_t* neg_meta_out_dev, void* stream) {
  TRS_REQUIRE((pos_meta_out_dev == nullptr) == (neg_meta_out_dev == nullptr), "trs_epoch_presort_meta: id outputs");
  TRS_REQUIRE(pos_dev && neg_dev && item_meta_dev && keys_dev && vals_dev && temp_dev && sorted_keys_out &&
                  sorted_vals_out, "trs_epoch_presort_meta: NULL argument");
  TRS_REQUIRE(n_batches > 0 && batch > 0 && M > 0 && m >= 0 && m < M && n_cat > 0, "trs_epoch_presort_meta: bad sizes");
  const int64_t n_pos = n_batches * batch;
  TRS_REQUIRE(2 * n_pos < ((int64_t)1 << 32), "trs_epoch_presort_meta: slice too long");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)(2 * n_pos);
  uint32_t* kin = (uint32_t*)keys_dev;
  RefPayload* vin = (RefPayload*)vals_dev;
  hipLaunchKernelGGL(meta_refs_kernel, dim3(trs_grid(n_pos, TRS_BLOCK)), dim3(TRS_BLOCK), 0, s, pos_dev, neg_dev, n_pos,
                     batch, item_meta_dev, M, m, n_cat, kin, vin, err_flag_dev, pos_meta_out_dev, neg_meta_out_dev);
  TRS_CHECK_LAUNCH("meta_refs_kernel");
  (void)temp_bytes;
  *sorted_keys_out = (void*)(kin + n);
  *sorted_vals_out = (void*)(vin + n);
  // the column's references grouped by the counting sort (categories = rows; keys interleaved in kin)
  return launch_group<1>((const int32_t*)kin, (const int32_t*)nullptr, n_batches, batch, n_cat, bits_for(2 * batch),
                         kin + n, vin + n, s, "trs_epoch_presort_meta");
}

// One launch of the sorted item update for a step (used by trs_train_steps_sgd's sorted mode).
int trs_launch_sorted_item_update(const trs_tables* tables, const void* keys_step, const void* vals_step, int key_bytes,
                                  int64_t batch, int64_t n_batches_bits_items, const float* gz, float lr,
                                  uint64_t* uown, uint32_t* udup, uint32_t stamp, const float* ustage,
                                  const int32_t* user_ids, hipStream_t s) {
  SortedArgs a = {};
  a.T = *tables;
  a.keys = keys_step;
  a.vals = (const RefPayload*)vals_step;
  a.user_ids = user_ids;
  a.B = batch;
  a.item_bits = (int)n_batches_bits_items;
  a.gz = gz;
  a.lr = lr;
  a.uown = uown;
  a.udup = udup;
  a.stamp = stamp;
  a.ustage = ustage;
  RowCfg c;
  if (!pick_row_cfg(tables->D, c)) {
    trs_set_error("unsupported n_factors D=%d", tables->D);
    return TRS_E_ARG;
  }
  const int tpw = TRS_WAVE / c.g;
  const dim3 gr(trs_grid((2 * batch + tpw - 1) / tpw, TRS_BLOCK / TRS_WAVE)), bl(TRS_BLOCK);
  const dim3 gs(trs_grid((2 * batch + RUN_CHUNK - 1) / RUN_CHUNK, 1));  // staged form: one 64-reference chunk per workgroup
#define TRS_SL(V, GG, KK, FULL)                                                                                      \
  {                                                                                                                  \
    if (key_bytes == 4 && ustage) hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint32_t, V, GG, KK, FULL>), gs, bl, 0, s, a); \
    else if (key_bytes == 4) hipLaunchKernelGGL((sorted_item_update_kernel<uint32_t, V, GG, KK, FULL, false>), gr, bl, 0, s, a);     \
    else if (ustage) hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint64_t, V, GG, KK, FULL>), gs, bl, 0, s, a);             \
    else hipLaunchKernelGGL((sorted_item_update_kernel<uint64_t, V, GG, KK, FULL, false>), gr, bl, 0, s, a);                         \
  }
#define TRS_CASE(V, GG, KK)                                                                  \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                \
    if (V * GG * KK == tables->D) TRS_SL(V, GG, KK, true) else TRS_SL(V, GG, KK, false)      \
    TRS_CHECK_LAUNCH("sorted_item_update_kernel");                                           \
    return TRS_OK;                                                                           \
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
#undef TRS_SL
  trs_set_error("internal: no kernel for D=%d", tables->D);
  return TRS_E_ARG;
}

int trs_item_bits_for(int64_t n_items) { return bits_for(n_items); }

// Sorted-run update of one metadata column: the staged item kernel pointed at that column's tables.  opt (adaptive
// rules): item_* / gacc* / cut_* are the COLUMN's state, accumulators and cut-run list; adaptive kernels exist for the
// shapes meta_stage_kernel handles (D = 4*G, G in {8,16,32,64}).
int trs_launch_sorted_meta_update(const trs_tables* tables, int m, float* lin_or_scratch, const void* keys_step,
                                  const void* vals_step, int64_t batch, const float* gz, float lr, const float* xstage,
                                  int64_t xpass, int fmsub, const OptArgs* opt, int parity, hipStream_t s) {
  SortedArgs a = {};
  a.T = *tables;
  a.keys = keys_step;
  a.vals = (const RefPayload*)vals_step;
  a.B = batch;
  a.item_bits = bits_for(tables->n_meta[m]);
  a.gz = gz;
  a.lr = lr;
  a.ustage = xstage;
  a.xpass = xpass;
  a.fmsub = fmsub;
  a.tab = tables->meta[m];
  a.tab_lin = lin_or_scratch;
  a.parity = parity & 1;
  const int kind = opt ? opt->kind : OPT_SGD;
  if (opt) a.o = *opt;
  RowCfg c;
  if (!pick_row_cfg(tables->D, c)) {
    trs_set_error("unsupported n_factors D=%d", tables->D);
    return TRS_E_ARG;
  }
  const dim3 gs(trs_grid((2 * batch + RUN_CHUNK - 1) / RUN_CHUNK, 1)), bl(TRS_BLOCK), gc(128);
  if (kind != OPT_SGD) {
    const trs_tables T = *tables;
#define TRS_CASE(GG)                                                                                                  \
  if (c.vec == 4 && c.g == GG && c.k == 1 && 4 * GG == tables->D) {                                                   \
    if (kind == OPT_ADAM) {                                                                                           \
      hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint32_t, 4, GG, 1, true, OPT_ADAM>), gs, bl, 0, s, a);     \
      hipLaunchKernelGGL((cut_rows_apply_kernel<4, GG, 1, true, OPT_ADAM>), gc, bl, 0, s, T, a.o, a.parity, a.tab,     \
                         a.tab_lin);                                                                                  \
    } else {                                                                                                          \
      hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint32_t, 4, GG, 1, true, OPT_ADAGRAD>), gs, bl, 0, s, a);  \
      hipLaunchKernelGGL((cut_rows_apply_kernel<4, GG, 1, true, OPT_ADAGRAD>), gc, bl, 0, s, T, a.o, a.parity, a.tab,  \
                         a.tab_lin);                                                                                  \
    }                                                                                                                 \
    TRS_CHECK_LAUNCH("sorted_item_update_staged_kernel");                                                             \
    return TRS_OK;                                                                                                    \
  }
    TRS_CASE(8)
    TRS_CASE(16)
    TRS_CASE(32)
    TRS_CASE(64)
#undef TRS_CASE
    trs_set_error("adaptive rules on metadata columns need D in {32, 64, 128, 256} (got %d)", tables->D);
    return TRS_E_ARG;
  }
#define TRS_CASE(V, GG, KK)                                                                                          \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                                        \
    if (V * GG * KK == tables->D)                                                                                    \
      hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint32_t, V, GG, KK, true>), gs, bl, 0, s, a);            \
    else                                                                                                             \
      hipLaunchKernelGGL((sorted_item_update_staged_kernel<uint32_t, V, GG, KK, false>), gs, bl, 0, s, a);           \
    TRS_CHECK_LAUNCH("sorted_item_update_staged_kernel");                                                            \
    return TRS_OK;                                                                                                   \
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
  trs_set_error("internal: no kernel for D=%d", tables->D);
  return TRS_E_ARG;
}

// Fused launch of the staged item update and the duplicated-user update (both with 32-bit keys).
int trs_launch_sorted_updates_fused(const trs_tables* tables, const void* keys_step, const void* vals_step,
                                    int64_t batch, int64_t item_bits, const float* gz, float lr, const float* ustage,
                                    const void* ukeys_step, const void* uvals_step, int64_t q0, const float* du,
                                    const OptArgs* opt, int parity, int64_t xpass, int fmsub, int skip_single,
                                    const uint8_t* uflags_step, const int32_t* user_ids_step, hipStream_t s) {
  SortedArgs ia = {};
  ia.skip_single = skip_single;
  ia.T = *tables;
  ia.keys = keys_step;
  ia.vals = (const RefPayload*)vals_step;
  ia.B = batch;
  ia.item_bits = (int)item_bits;
  ia.gz = gz;
  ia.lr = lr;
  ia.ustage = ustage;
  UserDupArgs ua = {};
  ua.T = *tables;
  ua.ukeys = ukeys_step;
  ua.uvals = (const uint32_t*)uvals_step;
  ua.B = batch;
  ua.q0 = q0;
  ua.user_bits = bits_for(tables->n_users);
  ua.du = du;
  ua.gz = gz;
  ua.lr = lr;
  ua.uflags = uflags_step;
  ua.user_ids = user_ids_step;
  const int kind = opt ? opt->kind : OPT_SGD;
  if (opt) {
    ia.o = *opt;
    ua.o = *opt;
  }
  ia.parity = parity & 1;
  ia.xpass = xpass;
  ia.fmsub = fmsub;
  RowCfg c;
  if (!pick_row_cfg(tables->D, c)) {
    trs_set_error("unsupported n_factors D=%d", tables->D);
    return TRS_E_ARG;
  }
  const int nu = trs_grid((batch + TRS_WAVE - 1) / TRS_WAVE, TRS_BLOCK / TRS_WAVE);
  const int ni = trs_grid((2 * batch + RUN_CHUNK - 1) / RUN_CHUNK, 1);
  const dim3 gr(nu + ni), bl(TRS_BLOCK), gc(64);
  const trs_tables T = *tables;
#define TRS_FUSED(V, GG, KK, FULL)                                                                                \
  {                                                                                                              \
    if (kind == OPT_ADAM) {                                                                                      \
      hipLaunchKernelGGL((sorted_updates_fused_kernel<V, GG, KK, FULL, OPT_ADAM>), gr, bl, 0, s, ia, ua, nu);     \
      hipLaunchKernelGGL((cut_rows_apply_kernel<V, GG, KK, FULL, OPT_ADAM>), gc, bl, 0, s, T, ia.o, ia.parity, (float*)nullptr, (float*)nullptr); \
    } else if (kind == OPT_ADAGRAD) {                                                                            \
      hipLaunchKernelGGL((sorted_updates_fused_kernel<V, GG, KK, FULL, OPT_ADAGRAD>), gr, bl, 0, s, ia, ua, nu);  \
      hipLaunchKernelGGL((cut_rows_apply_kernel<V, GG, KK, FULL, OPT_ADAGRAD>), gc, bl, 0, s, T, ia.o, ia.parity, (float*)nullptr, (float*)nullptr); \
    } else {                                                                                                     \
      hipLaunchKernelGGL((sorted_updates_fused_kernel<V, GG, KK, FULL>), gr, bl, 0, s, ia, ua, nu);               \
    }                                                                                                            \
  }
#define TRS_CASE(V, GG, KK)                                                                                      \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                                    \
    if (V * GG * KK == tables->D) TRS_FUSED(V, GG, KK, true) else TRS_FUSED(V, GG, KK, false)                     \
    TRS_CHECK_LAUNCH("sorted_updates_fused_kernel");                                                             \
    return TRS_OK;                                                                                               \
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
#undef TRS_FUSED
  trs_set_error("internal: no kernel for D=%d", tables->D);
  return TRS_E_ARG;
}

int trs_launch_sorted_user_dup_update(const trs_tables* tables, const void* ukeys_step, const void* uvals_step,
                                      int key_bytes, int64_t batch, int64_t q0, const float* du, const float* gz,
                                      float lr, hipStream_t s) {
  UserDupArgs a = {};
  a.T = *tables;
  a.ukeys = ukeys_step;
  a.uvals = (const uint32_t*)uvals_step;
  a.B = batch;
  a.q0 = q0;
  a.user_bits = bits_for(tables->n_users);
  a.du = du;
  a.gz = gz;
  a.lr = lr;
  RowCfg c;
  if (!pick_row_cfg(tables->D, c)) {
    trs_set_error("unsupported n_factors D=%d", tables->D);
    return TRS_E_ARG;
  }
  const dim3 gr(trs_grid((batch + TRS_WAVE - 1) / TRS_WAVE, TRS_BLOCK / TRS_WAVE)), bl(TRS_BLOCK);
#define TRS_UL(V, GG, KK, FULL)                                                                                  \
  {                                                                                                              \
    if (key_bytes == 4) hipLaunchKernelGGL((sorted_user_dup_update_kernel<uint32_t, V, GG, KK, FULL>), gr, bl, 0, s, a); \
    else hipLaunchKernelGGL((sorted_user_dup_update_kernel<uint64_t, V, GG, KK, FULL>), gr, bl, 0, s, a);         \
  }
#define TRS_CASE(V, GG, KK)                                                                  \
  if (c.vec == V && c.g == GG && c.k == KK) {                                                \
    if (V * GG * KK == tables->D) TRS_UL(V, GG, KK, true) else TRS_UL(V, GG, KK, false)      \
    TRS_CHECK_LAUNCH("sorted_user_dup_update_kernel");                                       \
    return TRS_OK;                                                                           \
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
#undef TRS_UL
  trs_set_error("internal: no kernel for D=%d", tables->D);
  return TRS_E_ARG;
}
