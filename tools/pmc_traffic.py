"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md prescribes) -> profiles/rNN_pmc_traffic.json, which bench.py reports as roofline.traffic.

    python tools/pmc_traffic.py --fetch <..counter_collection.csv> --write <..counter_collection.csv> --out profiles/r01_pmc_traffic.json

Correction (gfx950): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts a 128-B read request as 64 B, so
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import argparse
import csv
import json
import re
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"^trs::", "", name)
        name = re.sub(r"[<(].*$", "", name)  # template arguments / signature
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--source", default="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                                        "`python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-kernel-events`, "
                                        "config c2, MI355X")
    a = ap.parse_args()
    f, nf = per_kernel(a.fetch, "FETCH_SIZE")
    w, _ = per_kernel(a.write, "WRITE_SIZE")
    keep = [k for k in f if k in w and ("trs" in k or "kernel" in k) and "at::" not in k and "ROCPRIM_400001" not in k]
    out = {"source": a.source,
           "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH_SIZE counts 128-B "
                         "read requests as 64 B, MI355X_MICROARCH.md 'HBM'); the x2 is calibrated for 16-B-per-lane "
                         "reads",
           "kernels": {k: {"launches": nf[k], "FETCH_SIZE_KB_per_launch": f[k], "WRITE_SIZE_KB_per_launch": w[k],
                           "traffic_bytes_per_launch": (2 * f[k] + w[k]) * 1024} for k in sorted(keep)}}
    json.dump(out, open(a.out, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:60s} {v['traffic_bytes_per_launch'] / 1e6:10.2f} MB/launch over {v['launches']} launches")


if __name__ == "__main__":
    main()
