"""Per-kernel summary (calls, mean / median / min / max ns, total) and optional timeline of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace ... -d DIR` writes DIR/<host>/<pid>_results.db on ROCm 7.2).
usage: python tools/rocpd_stats.py DB [--csv OUT.csv] [--timeline OUT.txt [--from-kernel NAME]]"""
import re
import sqlite3
import statistics
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
    by = {}
    for name, s, e, *_ in rows:
        by.setdefault(short(name), []).append(e - s)
    lines = ["kernel,calls,mean_ns,median_ns,min_ns,max_ns,total_ns"]
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'"{k}",{len(v)},{sum(v) / len(v):.0f},{statistics.median(v):.0f},{min(v)},{max(v)},{sum(v)}')
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
    else:
        print("\n".join(lines[:40]))
    if "--timeline" in sys.argv:
        out = open(sys.argv[sys.argv.index("--timeline") + 1], "w")
        t0 = rows[0][1]
        last_end = {}
        prev_end = None
        for name, s, e, stream, gx, wx in rows:
            gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.2f}"
            out.write(f"{(s - t0) / 1e3:12.2f} us  dur {(e - s) / 1e3:9.2f} us  gap {gap:>8}  s{stream}  "
                      f"g{gx // max(wx, 1)}  {short(name)[:80]}\n")
            prev_end = max(prev_end or 0, e)
        out.close()


main()
