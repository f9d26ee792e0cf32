#!/usr/bin/env python3
"""Time the tile convolution (ct_gconv2d path) on GMFlow's shapes; CT_HIP_LIB selects a diagnostic library."""
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
for (cin, cout, kh, kw) in ((256, 192, 3, 3), (128, 256, 3, 3), (256, 128, 1, 5), (256, 128, 5, 1), (128, 128, 1, 5), (256, 126, 3, 3)):
    x = torch.randn(2, cin, 128, 224, device="cuda")
    wt = torch.randn(cout, cin, kh, kw, device="cuda") / (cin * kh * kw) ** 0.5
    wp, bp = ct_hip.pack_gconv_weight(wt, torch.randn(cout, device="cuda"))
    us = timeit(lambda: ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=2))
    fl = 2.0 * 2 * 128 * 224 * cin * cout * kh * kw
    print("conv %3d -> %3d %dx%d at 2x128x224: %7.1f us  %6.1f TFLOP/s f32-equivalent" % (cin, cout, kh, kw, us, fl / us / 1e6))
