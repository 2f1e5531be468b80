#!/usr/bin/env python
"""Measured HBM traffic (rocprofv3 PMC FETCH_SIZE / WRITE_SIZE, separate passes) next to the algorithmic bytes of
tools/pmc_probe.py's launches.  FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide coalesced read stream
(MI355X_MICROARCH.md, section HBM) -> corrected x2; WRITE_SIZE is reported as-is (uncalibrated).
usage: python tools/pmc_report.py gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db gpurun_out/pmc_fetch.log"""
import hashlib, json, os, re, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources the profiled launches are compiled from: bench.py refuses a committed profile whose digests differ from the tree it runs in
PROFILED_SOURCES = ["gemv.hip", "gemv_dot.h", "attn_decode.hip", "attn_decode_body.h", "granule.h", "tp_body.h", "dl_common.h"]


def source_digests():
    out = {}
    for f in PROFILED_SOURCES:
        with open(os.path.join(ROOT, "dynamic_llava_amd", "csrc", f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def per_dispatch(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, value, duration from counters_collection where counter_name=? order by start", (counter,)).fetchall()
    out = []
    for name, gx, gy, gz, wx, val, dur in rows:
        if "dl::" not in name:
            continue
        short = re.sub(r"^void ", "", name).split("(")[0]
        out.append((short, gx // max(wx, 1), gy, gz, val * 1024.0, dur))
    return out


def main():
    fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
    write = per_dispatch(sys.argv[2], "WRITE_SIZE")
    plan = json.loads(next(l for l in open(sys.argv[3]) if l.startswith("PLAN "))[5:])
    # the probe launches every case exactly 3x, in plan order: chunk the main-kernel dispatches by 3
    main = lambda rows: [r for r in rows if any(k in r[0] for k in ("attn_decode_split_kernel", "gemv_kernel", "gemv_b1_plain_kernel", "gemv_b1_plain_halves_kernel", "gemv_qkv_attn_kernel", "gemv_gu_tp_kernel", "linear_packed_kernel", "rmsnorm_kernel"))]
    chunk = lambda rows: [rows[i : i + 3] for i in range(0, len(rows), 3)]
    gf, gw = chunk(main(fetch)), chunk(main(write))
    assert len(gf) == len(plan) == len(gw), (len(gf), len(gw), len(plan))
    print(f"{'case':28s} {'kernel':34s} {'grid':>14s} {'algorithmic MB':>15s} {'FETCH x2 MB':>12s} {'WRITE MB':>9s} {'traffic/alg':>11s} {'us':>8s}")
    res = []
    for p, f, w in zip(plan, gf, gw):
        fb = min(x[4] for x in f) * 2.0
        wb = min(x[4] for x in w)
        us = min(x[5] for x in f) / 1e3
        alg = p["algorithmic_bytes"]
        print(f"{p['tag']:28s} {f[0][0][:34]:34s} {str(f[0][1:4]):>14s} {alg/1e6:15.2f} {fb/1e6:12.2f} {wb/1e6:9.3f} {(fb+wb)/alg:11.3f} {us:8.2f}")
        res.append({"case": p["tag"], "kernel": f[0][0], "algorithmic_bytes": alg, "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_over_algorithmic": round((fb + wb) / alg, 4), "kernel_us_under_pmc": round(us, 2)})
    res.append({"case": "_meta", "source_digests": source_digests(),
                "note": "sha256[:16] of the csrc files the profiled kernels are built from, at the time of the PMC passes; bench.py recomputes them and refuses this file if one differs"})
    print("JSON " + json.dumps(res))


if __name__ == "__main__":
    main()
