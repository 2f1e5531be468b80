#!/usr/bin/env python
"""Per-variant averages of every counter in rocprofv3 PMC databases of tools/pmc_linear_packed_probe.py (one database per counter group).
usage: python tools/pmc_linear_packed_report.py <probe stdout> <db> [<db> ...]"""
import collections, json, re, sqlite3, sys
variants = json.loads(next(l for l in open(sys.argv[1]) if l.startswith("VARIANTS "))[9:])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, grid_size_x, workgroup_size_x, counter_name, value, duration, start from counters_collection order by start").fetchall()
    cur, seen_warm = None, 0
    per_dispatch = collections.OrderedDict()
    for did, name, gx, wx, cname, val, dur, start in rows:
        d_ = per_dispatch.setdefault(did, dict(name=name, grid=gx // max(wx, 1), dur=dur, c={}))
        d_["c"][cname] = d_["c"].get(cname, 0.0) + val
    for d in per_dispatch.values():
        if "pack_x_tiles_kernel" in d["name"] and d["grid"] <= 4:  # a marker (the probe's own X packing has a larger grid)
            seen_warm += 1
            cur = seen_warm // 2 - 1 if seen_warm % 2 == 0 else None
            continue
        if cur is None or cur >= len(variants):
            continue
        if any(k in d["name"] for k in ("pack_weight", "fill", "randn", "distribution", "elementwise", "copy")):
            continue
        key = variants[cur]
        agg[key]["kernels"].append(re.sub(r"^void ", "", d["name"]).split("(")[0][:60])
        agg[key]["us"].append(d["dur"] / 1e3)
        for k, v in d["c"].items():
            agg[key][k].append(v)
for key in variants:
    d = agg.get(key)
    if not d:
        continue
    n = len(d["us"])
    kern = collections.Counter(d["kernels"]).most_common(2)
    parts = [f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(d.items()) if k not in ("kernels",)]
    print(f"{key:34s} ({n} dispatches; {', '.join(f'{a} x{b}' for a, b in kern)})\n    " + "  ".join(parts))
