#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max, like --stats CSV.
usage: python tools/prof_summary.py gpurun_out/prof1/bench_results.db [top_n] [--after KERNEL_SUBSTRING]
--after: only dispatches that START after the last dispatch whose name contains the substring (a marker kernel the traced script launches
once its set-up -- weight init, warm-up -- is over)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    args = list(sys.argv[1:])
    after = None
    if "--after" in args:
        i = args.index("--after")
        after = args[i + 1]
        del args[i : i + 2]
    db = args[0]
    top = int(args[1]) if len(args) > 1 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels").fetchall()
    if after is not None:
        marks = [e for name, s, e in rows if after in name]
        if not marks:
            raise SystemExit(f"no dispatch named *{after}* in {db}")
        rows = [r for r in rows if r[1] >= max(marks)]
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
