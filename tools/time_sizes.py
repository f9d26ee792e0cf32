#!/usr/bin/env python3
"""GPU box: per-kernel time of the fused Reinhard call vs image size (does the working set's cache residency matter?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, ct_hip
torch.cuda.set_device(0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for e in ev:
    e.record()
torch.cuda.synchronize()
for (h, w, B) in [(16, 16, 1), (64, 64, 8), (256, 256, 8), (540, 960, 1), (1080, 1920, 4), (1080, 1920, 2), (1080, 1920, 1), (540, 960, 4), (540, 960, 16), (270, 480, 64), (2160, 3840, 1), (2160, 3840, 2)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    T = torch.rand((B, h, w, 3), device="cuda", generator=g); R = torch.rand((B, h, w, 3), device="cuda", generator=g)
    out = torch.empty_like(T)
    for _ in range(30):
        ct_hip.reinhard(T, R, out=out)
    torch.cuda.synchronize()
    ct_hip.profile_events(ev)
    ts, ta = [], []
    for _ in range(30):
        ct_hip.reinhard(T, R, out=out)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3); ta.append(ev[2].elapsed_time(ev[3]) * 1e3)
    ct_hip.profile_events(None)
    mp = B * h * w / 1e6
    print("%4dx%4d B=%2d (%5.1f MB in)  stats %6.1f us = %5.2f ns/Mpx-pair... %5.2f us per Mpixel-image | apply %6.1f us = %5.2f us per Mpixel" % (
        h, w, B, 2 * mp * 12, np.median(ts), 0, np.median(ts) / (2 * mp), np.median(ta), np.median(ta) / mp))
