#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max, like --stats CSV.
usage: python tools_prof_summary.py gpurun_out/prof1/bench_results.db [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    print(f"# {db}: {len(rows)} dispatches, {len(agg)} kernels, sum of kernel time {tot/1e6:.3f} ms, trace span {(t1-t0)/1e6:.3f} ms")
    print(f"{'kernel':110s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:110s} {a[0]:8d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:9.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}")


if __name__ == "__main__":
    main()
