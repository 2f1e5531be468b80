import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
M, N, K = 170, 12288, 4096
w = torch.randn(N, K, device=dev, dtype=dt) / K**0.5
wp = ops.pack_weight_tiles(w)
x = torch.randn(M, K, device=dev, dtype=dt)
ref = F.linear(x.float(), w.float())
tiles = -(-M // 16); tpw = -(-tiles // 4)
xpad = torch.cat([x, x[-1:].expand(4 * tpw * 16 - M, K)])
xpk = xpad.view(4 * tpw, 16, K // 64, 2, 4, 8).permute(2, 0, 3, 4, 1, 5).contiguous()
torch.cuda.synchronize()
nu = 3
for fl in (0, 2, 2, 2, 1, 1, 3):
    print(f"flags={fl} ...", end="", flush=True)
    xin, mk = (xpk, (M, K)) if fl & 1 else (x, None)
    got = ops.linear_packed(xin, wp, N, units_per_workgroup=nu, _flags=fl, _mk=mk).float()
    torch.cuda.synchronize()
    e = (got - ref).abs() / ref.abs().max()
    bad = e > 2e-2
    cols = bad.any(dim=0).nonzero().flatten()
    wgs = sorted(set((cols // (16 * nu)).tolist()))
    print(f" max err {float(e.max()):.2e} bad wgs {len(wgs)} {wgs[:16]}", flush=True)
