#!/usr/bin/env python3
"""Two batches in flight: the statistics sweep (issue bound) of one batch beside the apply sweep (HBM bound) of another, on two
streams.  Usage: bench_reinhard_streams.py [pairs per call]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 1080, 1920
torch.manual_seed(0)
sets = []
for s in range(2):
    t, r, g = (torch.rand(B, H, W, 3, device="cuda") for _ in range(3))
    sets.append((t, r, g, torch.empty_like(t), torch.zeros(B, 2, dtype=torch.float64, device="cuda")))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 2
        if two:
            with torch.cuda.stream(streams[k]):
                t, r, g, o, p = sets[k]
                ct_hip.reinhard_psnr(t, r, g, out=o, psnr_out=p)
        else:
            t, r, g, o, p = sets[k]
            ct_hip.reinhard_psnr(t, r, g, out=o, psnr_out=p)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for two in (False, True, False, True):
    run(20, two)
    dt = run(200, two)
    print("%s: %.1f us per call of %d pairs = %.0f pairs/s" % ("two streams" if two else "one stream ", dt * 1e6, B, B / dt))
