#!/usr/bin/env python3
"""Time ct_local_corr_flow_f32 (the matcher's refinement correlation, 128 x 224 tokens, batch 2) on flows of different smoothness:
the tile form shares a 4 x 8 tile's window box in LDS; tiles whose box exceeds the LDS budget take the per-pixel form.
CT_HIP_LCF_TILE=0 times the per-pixel kernel on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
b, h, w = 2, 128, 224
t0, t1 = torch.randn(b, h * w, 128, device="cuda"), torch.randn(b, h * w, 128, device="cuda")
yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device="cuda"), torch.arange(w, dtype=torch.float32, device="cuda"), indexing="ij")
flows = {
    "constant (2.25, -3.5)": torch.stack([torch.full_like(xx, 2.25), torch.full_like(xx, -3.5)], 0)[None].repeat(b, 1, 1, 1),
    "smooth (gradient 0.05 px/px)": torch.stack([0.05 * xx - 0.02 * yy, 0.03 * yy + 0.01 * xx], 0)[None].repeat(b, 1, 1, 1),
    "smooth + noise 0.3 px": torch.stack([0.05 * xx, 0.03 * yy], 0)[None].repeat(b, 1, 1, 1) + 0.3 * torch.randn(b, 2, h, w, device="cuda"),
    "smooth + noise 1 px": torch.stack([0.05 * xx, 0.03 * yy], 0)[None].repeat(b, 1, 1, 1) + 1.0 * torch.randn(b, 2, h, w, device="cuda"),
    "random 5 px": 5 * torch.randn(b, 2, h, w, device="cuda"),
}
for name, flow in flows.items():
    flow = flow.contiguous()
    us = timeit(lambda: ct_hip.local_corr_flow(t0, t1, flow, 4))
    print("local_corr_flow, flow %-30s %7.1f us" % (name, us))
