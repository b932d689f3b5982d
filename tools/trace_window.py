# -*- coding: utf-8 -*-
"""Timeline of a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W`: the kernels of the timed window (the
last K step pairs before the instrumented / pass measurements), gaps between consecutive kernels, per-kernel totals.
Usage: python tools/trace_window.py <kernel_trace.csv> [K]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "")) for r in rows))
k1 = [i for i, k in enumerate(ks) if "fwd_stage_kernel" in k[2]]
print("kernels", len(ks), "K1 launches", len(k1))
# the timed window = K consecutive K1 launches after the first W (warm-up): print from the W-th K1 on
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
first, last = k1[W], k1[W + K - 1]
t0 = ks[first][0]
prev_end = None
for i in range(max(first - 3, 0), min(last + 4, len(ks))):
    s, e, n, st = ks[i]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {n}")
    prev_end = max(prev_end or e, e)
print("window first K1 start -> last kernel end:", (max(k[1] for k in ks[first:last + 2]) - t0) / 1e3, "us")
