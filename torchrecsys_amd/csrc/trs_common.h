// trs_common.h — shared host/device helpers of libtrs_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/trs.h"

#define TRS_WAVE 64
#define TRS_BLOCK 256

// ------------------------------------------------------------------------------------------ errors
void trs_set_error(const char* fmt, ...);

#define TRS_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      trs_set_error(__VA_ARGS__);     \
      return TRS_E_ARG;               \
    }                                 \
  } while (0)

#define TRS_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      trs_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TRS_E_LAUNCH;                                                 \
    }                                                                      \
  } while (0)

// Tuning / A-B knobs of the launch paths (kernel selection, launch shapes).  The defaults are the measured best; nothing
// in the product path needs them.  Read from the TRS_* environment ONCE, when the library is first used (api.cpp) — no
// launch path calls getenv — and changeable afterwards through trs_tuning_set (include/trs.h: tests, tools/).
// -1 = "not set: decide by shape" where a knob overrides an automatic choice.
struct TrsTuning {
  int64_t grid_cap;       // TRS_GRID_CAP        workgroups of a grid-stride kernel (256 CUs x 16)
  int64_t pass_grid_cap;  // TRS_PASS_GRID_CAP   ... of pair_scores_kernel (4096)
  int64_t presort_grid_cap;  // TRS_PRESORT_GRID_CAP  ... of epoch_refs_kernel on the side stream (512: see presort.hip)
  int k1_iters;           // TRS_K1_ITERS        pipelined iterations per lane group of fwd_stage_kernel (0: by batch)
  int pass_iters;         // TRS_PASS_ITERS      ... of pair_scores_kernel (0: by batch)
  int pass_nt;            // TRS_PASS_NT         nontemporal rows in the scoring pass: bit 0 user, bit 1 item (-1)
  int k1_nt;              // TRS_K1_NT           nontemporal user rows in the one-launch step: 1 loads, 3 loads + stores (-1)
  int k1_wgs_per_cu;      // TRS_K1_WGS_PER_CU   workgroups per CU the one-launch step may count on being resident (2)
  int gemm32_no_glds;     // TRS_GEMM32_NO_GLDS  1: fp32 NT GEMMs on the register-staged 128 x 128 kernel
  int gemm16_tn_wide;     // TRS_GEMM16_TN_WIDE  0 | 1: weight-gradient GEMMs on 256 x 256 tiles (-1)
  int gemm16_tile;        // TRS_GEMM16_TILE     128 | 256 | 512: force the bf16-resident tile (0)
  int gemm16_no_glds;     // TRS_GEMM16_NO_GLDS  1: bf16 NT GEMMs on the register-staged 256 x 256 kernel
  int bn_final_two_sweeps;  // TRS_BN_FINAL_TWO_SWEEPS  1: the two-sweep finalise kernels
};
TrsTuning& trs_tuning();

// grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 16 blocks, no more (measured on the
// update kernels: 512 blocks 31 us, 1024 23 us, 4096 21 us, 16384 21 us).
static inline int trs_grid(int64_t work_items, int items_per_block) {
  const int64_t cap = trs_tuning().grid_cap;
  int64_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ------------------------------------------------------------------------------------------ ids
__device__ __forceinline__ int64_t trs_ld_idx(const void* p, int idx_bytes, int64_t t) {
  return idx_bytes == 8 ? ((const int64_t*)p)[t] : (int64_t)((const int32_t*)p)[t];
}
__device__ __forceinline__ void trs_st_idx(void* p, int idx_bytes, int64_t t, int64_t v) {
  if (idx_bytes == 8)
    ((int64_t*)p)[t] = v;
  else
    ((int32_t*)p)[t] = (int32_t)v;
}

// ------------------------------------------------------------------------------------------ Philox4x32-10
// Counter-based generator (Salmon et al., SC'11).  Restated bit-for-bit in oracle/loader.py.
struct trs_u4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ trs_u4 trs_philox4x32_10(uint64_t counter, uint64_t key) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  trs_u4 o = {c0, c1, c2, c3};
  return o;
}

// 64x64 -> high 64 bits
__host__ __device__ __forceinline__ uint64_t trs_mulhi64(uint64_t a, uint64_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// neg uniform over {0..n_items-1} \ {pos}: r uniform over n_items-1 values, shifted past pos.
__host__ __device__ __forceinline__ int64_t trs_sample_one_neg(uint64_t seed, uint64_t ctr, int64_t pos,
                                                               int64_t n_items) {
  trs_u4 r = trs_philox4x32_10(ctr, seed);
  uint64_t x = ((uint64_t)r.y << 32) | (uint64_t)r.x;
  if (n_items <= 1) return 0;  // degenerate: the reference would loop forever (dataset/dataset.py:443)
  int64_t v = (int64_t)trs_mulhi64(x, (uint64_t)(n_items - 1));
  return v + (v >= pos ? 1 : 0);
}

// Pairwise loss of one triple from its two scores: its value and d/d(neg) (d/d(pos) = -d/d(neg)), before the 1/B of the
// mean.  TRS_LOSS_HINGE: the reference's clamp(neg - pos + 1, min=0) (helper/loss.py:5-9; torch's clamp passes the
// gradient at 0).  TRS_LOSS_BPR (BASELINE.json north_star; not in the reference): -log sigmoid(pos - neg) = softplus(neg - pos).
__device__ __forceinline__ void trs_pair_loss(int kind, float sp, float sn, float& value, float& dneg) {
  if (kind == TRS_LOSS_BPR) {
    const float x = sn - sp;
    value = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
    dneg = 1.0f / (1.0f + expf(-x));
  } else {
    const float h = sn - sp + 1.0f;
    value = fmaxf(h, 0.f);
    dneg = h >= 0.f ? 1.f : 0.f;
  }
}

// Sampler with the options of trs_sampler (include/trs.h); S.max_tries == 0 means "no options" (plain sampler).
struct TrsSampler {  // trs_sampler by value, for kernel arguments
  int k_neg, popularity, max_tries;
  const int64_t* seen_off;
  const int32_t* seen_items;
  const int32_t* pop_items;
  int64_t pop_n;
  int64_t seen_users;  // rows of the seen CSR (seen_off has seen_users + 1 entries)
};
static inline TrsSampler trs_sampler_args(const trs_sampler* s) {
  TrsSampler r = {1, 0, 0, nullptr, nullptr, nullptr, 0, 0};
  if (s) {
    r.k_neg = s->k_neg < 1 ? 1 : s->k_neg;
    r.popularity = s->popularity;
    r.max_tries = s->max_tries < 1 ? 1 : s->max_tries;
    r.seen_off = s->seen_off;
    r.seen_items = s->seen_items;
    r.pop_items = s->pop_items;
    r.pop_n = s->pop_n;
    r.seen_users = s->seen_users;
  }
  return r;
}
__device__ __forceinline__ bool trs_user_has_item(const TrsSampler& S, int64_t u, int64_t item) {
  // a user id outside the CSR (reported by the scorer's id check later) has seen nothing: no read beyond seen_off
  if ((uint64_t)u >= (uint64_t)S.seen_users) return false;
  int64_t lo = S.seen_off[u], hi = S.seen_off[u + 1];
  while (lo < hi) {  // binary search in the user's sorted positives
    const int64_t mid = (lo + hi) >> 1;
    const int64_t v = S.seen_items[mid];
    if (v == item) return true;
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return false;
}
__device__ __forceinline__ int64_t trs_sample_neg_opt(uint64_t seed, uint64_t ctr, int64_t u, int64_t pos,
                                                      int64_t n_items, const TrsSampler& S) {
  if (S.max_tries == 0) return trs_sample_one_neg(seed, ctr, pos, n_items);
  if (n_items <= 1) return 0;
  int64_t c = 0;
  for (int k = 0; k < S.max_tries; ++k) {
    const trs_u4 r = trs_philox4x32_10(ctr, seed + (uint64_t)k * 0x9E3779B97F4A7C15ull);
    const uint64_t x = ((uint64_t)r.y << 32) | (uint64_t)r.x;
    const int64_t v = (int64_t)trs_mulhi64(x, (uint64_t)(n_items - 1));
    c = v + (v >= pos ? 1 : 0);  // uniform over the items other than the positive
    if (S.popularity) {
      const uint64_t x2 = ((uint64_t)r.w << 32) | (uint64_t)r.z;
      const int64_t cp = S.pop_items[(int64_t)trs_mulhi64(x2, (uint64_t)S.pop_n)];
      if (cp == pos) continue;  // the row's own positive: next candidate (the uniform c stays as the fallback)
      c = cp;
    }
    if (!S.seen_off || !trs_user_has_item(S, u, c)) return c;
  }
  return c;
}

// ------------------------------------------------------------------------------------------ Feistel shuffle
// Keyed bijection of [0,N): 4-round balanced Feistel network on 2*hb bits (2*hb >= ceil(log2 N)), cycle-walked back
// into range.  Restated in oracle/loader.py::feistel_perm.
__host__ __device__ __forceinline__ uint32_t trs_mix32(uint32_t x, uint32_t k) {
  x ^= k;
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ int trs_feistel_half_bits(int64_t N) {
  int bits = 1;
  while (bits < 63 && ((int64_t)1 << bits) < N) ++bits;
  return (bits + 1) / 2;
}
__host__ __device__ __forceinline__ int64_t trs_feistel_perm(int64_t q, int64_t N, uint64_t key, int hb) {
  if (key == 0) return q;
  const uint64_t mask = ((uint64_t)1 << hb) - 1;
  uint64_t x = (uint64_t)q;
  do {
    uint32_t L = (uint32_t)(x >> hb), R = (uint32_t)(x & mask);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t rk = (uint32_t)(key >> (16 * (r & 3))) ^ (uint32_t)(key >> 32) ^ (0xA511E9B3u * (uint32_t)(r + 1));
      uint32_t F = trs_mix32(R, rk) & (uint32_t)mask;
      uint32_t nL = R;
      R = L ^ F;
      L = nL;
    }
    x = ((uint64_t)L << hb) | (uint64_t)R;
  } while (x >= (uint64_t)N);
  return (int64_t)x;
}

// ------------------------------------------------------------------------------------------ reductions
// sum over the G lanes of an aligned lane group (G power of two <= 64); every lane of the group gets the sum.
template <int G>
__device__ __forceinline__ float trs_group_sum(float v) {
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float trs_wave_sum(float v) { return trs_group_sum<64>(v); }
__device__ __forceinline__ int trs_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
