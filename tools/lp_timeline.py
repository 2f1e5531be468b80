#!/usr/bin/env python
"""Per-wave timeline of one dl_linear_packed launch (s_memtime stamps written by the kernel when the measurement hook is set): entry -> first
ring step landed -> k loop done -> hand-over done -> stores done, as medians / maxima over workgroups, per role.
  python tools/lp_timeline.py --shape qkv --nu 6 --ks 2"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="qkv"); ap.add_argument("--nu", type=int, default=3); ap.add_argument("--ks", type=int, default=1); ap.add_argument("--m", type=int, default=170)
ap.add_argument("--rowx", action="store_true"); ap.add_argument("--silu", action="store_true")
a = ap.parse_args()
H, I = 4096, 11008
N, K = {"qkv": (3 * H, H), "o": (H, H), "gate|up": (2 * I, H), "down": (H, I)}[a.shape]
dev, dt = "cuda", torch.bfloat16
ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(4)]
wps = [ops.pack_weight_tiles(w, gate_up_pairs=a.silu) for w in ws]
EPI = ops.LP_SILU_PAIR if a.silu else ops.LP_STORE
x = torch.randn(a.m, K, device=dev, dtype=dt); xpk = ops.pack_x_tiles(x)
wsb = ops.linear_packed_workspace(a.m, N, K, dev, 0, a.nu, a.ks)
n_wg = -(-(N // 16) // a.nu) * a.ks
st = torch.zeros(n_wg * 10 * 8, dtype=torch.int64, device=dev)
xin, mk = (x, None) if a.rowx else (xpk, (a.m, K))
for wp in wps: ops.linear_packed(xin, wp, N, x_packed_mk=mk, units_per_workgroup=a.nu, k_split=a.ks, workspace=wsb, epilogue=EPI)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.linear_packed(xin, wps[1], N, x_packed_mk=mk, units_per_workgroup=a.nu, k_split=a.ks, workspace=wsb, epilogue=EPI, stamps=st)
e1.record()
torch.cuda.synchronize()
launch_us = e0.elapsed_time(e1) * 1e3
s = st.view(n_wg, 10, 8).cpu().double()[:, :6]
# s_memtime counters of different XCDs are not synchronised: every wave is reported RELATIVE TO ITS OWN ENTRY stamp; ticks are converted with the
# launch's event time over the longest entry -> end span (the shader clock under this load)
span = (s[:, 2:, 4] - s[:, 2:, 0])
us_per_tick = launch_us / float(span.max())
n_sets = n_wg // a.ks
print(f"{a.shape} M={a.m} nu={a.nu} ks={a.ks} workgroups={n_wg}: launch {launch_us:.1f} us between events (includes ~5 us of launch overhead); 1 tick = {us_per_tick * 1e3:.3f} ns ({1e-3 / us_per_tick:.2f} GHz)")
for role, sel in (("partner", torch.arange(n_wg) < n_sets * (a.ks - 1)), ("reducer/last", torch.arange(n_wg) >= n_sets * (a.ks - 1))):
    if sel.sum() == 0: continue
    r = s[sel]
    for wname, wsel in (("loader", slice(0, 2)), ("consumer", slice(2, 6))):
        rr = r[:, wsel, :]
        line = f"  {role:13s} {wname:8s} (us after the wave's own entry):"
        for k, lab in enumerate(("entry", "first step landed", "k loop done", "hand-over done", "stores done")):
            if k == 0: continue
            ok = rr[:, :, k] > 0
            if not ok.any(): continue
            v = ((rr[:, :, k] - rr[:, :, 0]) * us_per_tick)[ok]
            line += f" {lab} med {float(v.median()):6.2f} max {float(v.max()):6.2f} |"
        print(line)
