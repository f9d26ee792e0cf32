#!/usr/bin/env python3
"""Time GMFlow.forward as DMSCT calls it (bidir + occlusion) at HxW (default 540x960 -> inference 512x896)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
from unimatch import GMFlow
from oracle.gmflow import derive_matcher_inference_size   # host arithmetic only (bench helper)

H = int(sys.argv[1]) if len(sys.argv) > 1 else 540
W = int(sys.argv[2]) if len(sys.argv) > 2 else 960
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
m = GMFlow().cuda()
a, b = torch.rand(1, 3, H, W, device="cuda") * 255, torch.rand(1, 3, H, W, device="cuda") * 255
size = derive_matcher_inference_size((1, 3, H, W))
for _ in range(2):
    m(a, b, inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m(a, b, inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flop = 3.58357106688e12 * (size[0] * size[1]) / (512 * 896)
print("GMFlow %dx%d (inference %s): %.2f ms/pair, %.2f pairs/s, %.1f TFLOP/s (%.1f%% of 157.3 TF fp32 MFMA)" % (
    H, W, size, dt * 1e3, 1 / dt, flop / dt / 1e12, flop / dt / 157.3e12 * 100))
