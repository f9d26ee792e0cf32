#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import methods.linear as lin
B = 4
rng = np.random.default_rng(0)
t = torch.from_numpy(rng.random((B, 1080, 1920, 3), dtype=np.float32)).cuda()
r = torch.from_numpy(rng.random((B, 1080, 1920, 3), dtype=np.float32)).cuda()
out = torch.empty_like(t)
for _ in range(3):
    lin.monge_kantorovitch_color_transfer_cuda(t, r, out_dtype=torch.float32, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    lin.monge_kantorovitch_color_transfer_cuda(t, r, out_dtype=torch.float32, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n / B
print("MK 1080p f32->f32: %.1f us/pair, %.0f pairs/s, %.2f TB/s (3 planes)" % (dt * 1e6, 1 / dt, 74649600 / dt / 1e12))
