#!/usr/bin/env python3
"""Time DCMCS3DI.forward(inference=True) at HxW (default 512x512), random init, on cuda:0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
from methods.dcmcs3di import DCMCS3DI

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
m = DCMCS3DI().cuda().eval()
left, right = torch.rand(1, 3, H, W, device="cuda"), torch.rand(1, 3, H, W, device="cuda")
for _ in range(2):
    m(left, right, inference=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m(left, right, inference=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flop = H * W * (6591040 + 390 * W)
print("DCMCS3DI %dx%d: %.3f ms/pair, %.2f pairs/s, %.1f TFLOP/s (%.1f%% of 157.3 TF fp32 MFMA)" % (
    H, W, dt * 1e3, 1 / dt, flop / dt / 1e12, flop / dt / 157.3e12 * 100))
