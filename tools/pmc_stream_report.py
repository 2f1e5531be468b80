#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 PMC database (tools/pmc_stream_probe.py).  usage: pmc_stream_report.py <db> [<db> ...]"""
import re, sqlite3, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for name, gx, wx, cname, val, dur in c.execute("select kernel_name, grid_size_x, workgroup_size_x, counter_name, value, duration from counters_collection"):
        if "dl::gemv" not in name and "gemm_smallm_staged" not in name:
            continue
        short = re.sub(r"^void ", "", name).split("(")[0].replace("dl::", "")[:48]
        key = (short, gx // max(wx, 1))
        agg[key][cname].append(val)
        agg[key]["us"].append(dur / 1e3)
for key in sorted(agg):
    d = agg[key]
    parts = [f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(d.items())]
    print(f"{key[0]:50s} grid {key[1]:5d}: " + "  ".join(parts))
