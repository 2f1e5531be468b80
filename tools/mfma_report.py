#!/usr/bin/env python
"""MFMA utilisation of the prefill's GEMM-shaped kernels from a rocprofv3 PMC pass over tools/prefill_kernels.py:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma -o m -- python tools/prefill_kernels.py
    python tools/mfma_report.py gpurun_out/pmc_mfma/m_results.db
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs) per dispatch: rocprofv3 reports both counters SUMMED
over their instances, and GRBM_GUI_ACTIVE has one instance per XCD (checked: value / kernel duration = 8 x 2.35 GHz), so the gfx94x
derived-metric formula needs the / 8.
Achieved TFLOP/s for the decoder GEMMs is computed from their known shapes (M = 170 / 631 rows of the bench prompt)."""
import collections, json, re, sqlite3, sys

N_CU, N_SIMD, N_XCD = 256, 4, 8
PEAK_TF = 2500.0


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    disp = collections.defaultdict(dict)
    for did, name, cn, val, dur in rows:
        d = disp[did]
        d["name"], d["dur"] = name, dur
        d[cn] = d.get(cn, 0.0) + val
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in disp.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        nm = re.sub(r"^void ", "", d["name"]).split("(")[0]
        a = agg[nm[:96]]
        a[0] += 1
        a[1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
        a[2] += d["GRBM_GUI_ACTIVE"]
        a[3] += d["dur"]
    out = []
    print(f"{'kernel':98s} {'calls':>6s} {'avg us':>8s} {'MfmaUtil %':>10s}")
    for nm, (n, busy, act, dur) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        if busy <= 0:
            continue
        util = 100.0 * busy / (act / N_XCD * N_CU * N_SIMD) if act else 0.0
        print(f"{nm:98s} {n:6d} {dur / n / 1e3:8.2f} {util:10.1f}")
        out.append({"kernel": nm, "calls": n, "avg_us_under_pmc": round(dur / n / 1e3, 2), "mfma_util_pct": round(util, 1)})
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
