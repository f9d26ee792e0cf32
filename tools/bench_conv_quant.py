#!/usr/bin/env python3
"""Tile quantisation of the tile convolution: time per launch against the number of (tile, group) units and the 512 resident
workgroups (tools/bench_conv_tile.py has the fixed shapes).  Usage: bench_conv_quant.py [cin cout kh kw]"""
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
cin, cout, kh, kw = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 128, 1, 5)
wt = torch.randn(cout, cin, kh, kw, device="cuda") / (cin * kh * kw) ** 0.5
wp, bp = ct_hip.pack_gconv_weight(wt, torch.randn(cout, device="cuda"))
for (h, w) in ((64, 224), (96, 224), (112, 224), (120, 224), (128, 224), (136, 224), (144, 224), (192, 224), (256, 224), (264, 224), (128, 256), (136, 240), (272, 480), (68, 120)):
    x = torch.randn(2, cin, h, w, device="cuda")
    us = timeit(lambda: ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=2))
    units = 2 * ((h + 7) // 8) * ((w + 31) // 32) * ((cout + 63) // 64)
    fl = 2.0 * 2 * h * w * cin * cout * kh * kw
    print("conv %3d -> %3d %dx%d at 2x%dx%d: %5d units  %7.1f us  %6.3f us/unit  %6.1f TFLOP/s f32-equivalent" % (cin, cout, kh, kw, h, w, units, us, us / units, fl / us / 1e6))
