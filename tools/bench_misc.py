#!/usr/bin/env python3
"""Informational rates of the section-8f kernels at 1080p: regrain / automated_color_grading, the 31 distortions, SSIM, iCID."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import ct_hip
from methods import iterative as it

def t_ms(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
t = torch.rand(H, W, 3, generator=g, dtype=torch.float64).cuda(); r = torch.rand(H, W, 3, generator=g, dtype=torch.float64).cuda()
rots = it.draw_rotations(4, seed=0)
graded = it.iterative_distribution_transfer_cuda(t, r, rotations=rots)
ms = t_ms(lambda: ct_hip.regrain(t, graded))
planes = 3 * 8 * H * W
print("regrain 1080p float64: %.2f ms/frame (%.1f frames/s)" % (ms, 1e3 / ms))
ms = t_ms(lambda: it.automated_color_grading_cuda(t, r, rotations=rots))
print("automated_color_grading 1080p float64: %.2f ms/pair (%.1f pairs/s)" % (ms, 1e3 / ms))
u8 = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).cuda()
for kind, p in (("brightness", 1.2), ("contrast", 0.8), ("saturation", 1.3), ("hue", 0.1), ("gamma", 1.2)):
    ms = t_ms(lambda: ct_hip.distort_u8(u8, kind, p), 20)
    print("distort %-10s 1080p u8 -> u8 + float32: %.3f ms (%.2f TB/s over 3 u8 planes in + 3 float32 planes out)" % (kind, ms, 3 * H * W * 5 / ms / 1e9))
