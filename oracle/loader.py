# -*- coding: utf-8 -*-
"""Index streams of the path: split, shuffle, negative samplers (integer work -> bit-exact class).
TEST INFRASTRUCTURE.

Reference files restated: dataset/dataset.py:56-64 (static negatives), :237-249 (split), :364-373 (shuffle),
:414-458 (batch slicing + dynamic sampler).  Third-party algorithms: sklearn.model_selection.train_test_split
(ShuffleSplit: RandomState(42).permutation, test = first ceil(test_size*N) entries), numpy legacy RandomState.randint,
torch.randperm (CPU generator).  Device-side streams (Philox4x32-10 sampler, Feistel shuffle) restate
torchrecsys_amd/csrc/trs_common.h bit for bit.
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------ reference host streams
def split_indices(N, split_ratio):
    """Row positions of (train, test) as sklearn's train_test_split(df, test_size=1-split_ratio, random_state=42)
    selects them (dataset/dataset.py:239-240).  SURVEY §3.1: verified equal to sklearn for several N."""
    test_size = 1 - split_ratio
    n_test = int(math.ceil(test_size * N))
    perm = np.random.RandomState(42).permutation(N)
    return perm[n_test:], perm[:n_test]


def static_negatives(N, num_items):
    """dataset/dataset.py:56-64: ONE np.random.randint(0, num_items, size=N) from the global legacy stream, no
    rejection of neg == pos, drawn before the split."""
    return np.random.randint(low=0, high=num_items, size=N)


def dynamic_negatives_walk(pos_item_ids, n_items, rng=np.random):
    """dataset/dataset.py:435-447, restated as a walk over the legacy randint stream: scalar
    `randint(0, n)` calls consume the same stream as `randint(0, n, size=k)`; each row takes the next stream value that
    differs from its own positive.  Consumes exactly the reference's number of draws."""
    pos = np.asarray(pos_item_ids, dtype=np.int64)
    B = pos.size
    out = np.empty(B, dtype=np.int64)
    k = 0
    while k < B:
        draws = rng.randint(0, n_items, size=B - k)
        # rows k.. take draws in order; a collision consumes a draw without finishing its row
        j = 0
        while j < draws.size and k < B:
            if draws[j] != pos[k]:
                out[k] = draws[j]
                k += 1
            j += 1
        # every draw of this block was consumed (j == draws.size) unless all rows finished exactly at its end
    return out


def dynamic_negatives_loop(pos_item_ids, n_items, rng=np.random):
    """The literal per-row loop (for small cases; pins the walk restatement)."""
    out = []
    for p in pos_item_ids:
        neg = rng.randint(0, n_items)
        while neg == p:
            neg = rng.randint(0, n_items)
        out.append(neg)
    return np.asarray(out, dtype=np.int64)


def batches_of(n_rows, batch_size):
    """[start, end) of every batch of FastDataLoader (last one partial, no drop_last) dataset/dataset.py:414-418,456."""
    return [(i, min(i + batch_size, n_rows)) for i in range(0, n_rows, batch_size)]


# ------------------------------------------------------------------------------------ device streams
M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """Philox4x32-10 of trs_common.h: counter words (c_lo, c_hi, 0, 0), key words (k_lo, k_hi).  Vectorised over
    `counter` (uint64 array).  Returns four uint32 arrays."""
    counter = np.asarray(counter, dtype=np.uint64)
    c0 = counter & M32
    c1 = counter >> np.uint64(32)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = np.uint64(int(key) & 0xFFFFFFFF)
    k1 = np.uint64((int(key) >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def _mulhi64(x, n):
    return np.array([(int(a) * int(n)) >> 64 for a in np.asarray(x, dtype=np.uint64).reshape(-1)], dtype=np.int64)


def device_negatives(pos, n_items, seed, offset):
    """trs_sample_neg / trs_batch_prepare: neg[t] = r + (r >= pos[t]), r = mulhi64(x, n_items-1),
    x = philox(offset + t, seed) words (y << 32 | x).  Uniform over the n_items-1 items other than the positive — the
    distribution of the reference's rejection loop."""
    pos = np.asarray(pos, dtype=np.int64)
    ctr = (np.arange(pos.size, dtype=np.uint64) + np.uint64(offset))
    x, y, _, _ = philox4x32_10(ctr, seed)
    r64 = (y.astype(np.uint64) << np.uint64(32)) | x.astype(np.uint64)
    v = _mulhi64(r64, n_items - 1)
    return v + (v >= pos)


def device_negatives_opt(users, pos, n_items, seed, offset, popularity=False, seen=None, pop_items=None, max_tries=8):
    """trs_sample_neg_opt (csrc/trs_common.h): the sampler with the options of trs_sampler (SURVEY 8f-4 — the reference
    itself only rejects the row's own positive, dataset/dataset.py:435-447).  Candidate of try k: Philox(counter,
    key = seed + k * 0x9E3779B97F4A7C15): words (x, y) -> uniform over the items other than the positive, words (z, w)
    -> popularity draw pop_items[mulhi(zw, len)] (a candidate equal to the positive is skipped); rejected while the
    user's positives (`seen`: dict user -> set, or CSR (offsets, items)) contain it; the last candidate is kept."""
    users, pos = np.asarray(users, dtype=np.int64), np.asarray(pos, dtype=np.int64)
    if isinstance(seen, tuple):
        off, items = (np.asarray(a) for a in seen)
        seen = {int(u): set(items[off[u]:off[u + 1]].tolist()) for u in np.unique(users)}
    out = np.zeros(pos.size, dtype=np.int64)
    G = 0x9E3779B97F4A7C15
    for t in range(pos.size):
        c = 0
        for k in range(max_tries):
            x, y, z, w = philox4x32_10(np.array([offset + t], dtype=np.uint64), (int(seed) + k * G) & 0xFFFFFFFFFFFFFFFF)
            v = ((int(y[0]) << 32 | int(x[0])) * (n_items - 1)) >> 64
            c = v + (1 if v >= pos[t] else 0)
            if popularity:
                cp = int(pop_items[((int(w[0]) << 32 | int(z[0])) * len(pop_items)) >> 64])
                if cp == pos[t]:
                    continue
                c = cp
            if seen is None or c not in seen.get(int(users[t]), ()):
                break
        out[t] = c
    return out


def _mix32(x, k):
    x = (x ^ k) & 0xFFFFFFFF
    x = (x * 0x9E3779B1) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE3D) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def feistel_half_bits(N):
    bits = 1
    while bits < 63 and (1 << bits) < N:
        bits += 1
    return (bits + 1) // 2


def feistel_perm(q, N, key):
    """trs_feistel_perm: keyed bijection of [0,N) (pure-Python ints; small cases)."""
    if key == 0:
        return q
    hb = feistel_half_bits(N)
    mask = (1 << hb) - 1
    x = q
    while True:
        L, R = x >> hb, x & mask
        for r in range(4):
            rk = ((key >> (16 * (r & 3))) & 0xFFFFFFFF) ^ ((key >> 32) & 0xFFFFFFFF) ^ ((0xA511E9B3 * (r + 1)) & 0xFFFFFFFF)
            F = _mix32(R, rk) & mask
            L, R = R, L ^ F
        x = (L << hb) | R
        if x < N:
            return x


def device_batch(stream_user, stream_item, neg_static, shuffle_key, t0, B, n_items, seed, offset, item_meta=None,
                 sampler=None):
    """trs_batch_prepare restated.  Returns dict user/pos/neg (+ pos_meta/neg_meta).  sampler: None or a dict
    k / popularity / seen / max_tries (trs_sampler): k > 1 = k * N epoch positions, position q -> row perm(q) % N."""
    N = len(stream_user)
    k = (sampler or {}).get("k", 1)
    rows = np.array([feistel_perm(t0 + t, N * k, shuffle_key) % N for t in range(B)], dtype=np.int64)
    u = np.asarray(stream_user)[rows].astype(np.int64)
    p = np.asarray(stream_item)[rows].astype(np.int64)
    if neg_static is not None:
        n = np.asarray(neg_static)[rows].astype(np.int64)
    elif sampler:
        n = device_negatives_opt(u, p, n_items, seed, offset, popularity=sampler.get("popularity", False),
                                 seen=sampler.get("seen"), pop_items=np.asarray(stream_item),
                                 max_tries=sampler.get("max_tries", 8))
    else:
        n = device_negatives(p, n_items, seed, offset)
    out = {"user": u, "pos": p, "neg": n}
    if item_meta is not None:
        out["pos_meta"] = np.asarray(item_meta)[p]
        out["neg_meta"] = np.asarray(item_meta)[n]
    return out
