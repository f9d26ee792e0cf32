#!/usr/bin/env python3
"""3x3 64->64 convolution on [2,64,512,512] (the DCMCS3DI ResB conv): exact-f32 MFMA kernel vs split-bf16 kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
import torch
import ct_hip as hip
shapes = [(2, 64, 64, 512, 512, 3), (2, 384, 128, 128, 224, 3), (2, 64, 64, 512, 512, 1)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:7])]
for (n, cin, cout, h, w, k) in shapes:
    x = torch.randn(n, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    b = torch.randn(cout, device="cuda"); res = torch.randn(n, cout, h, w, device="cuda")
    wp, bp = hip.pack_gconv_weight(wt, b)
    out = torch.empty(n, cout, h, w, device="cuda")
    flop = 2.0 * n * h * w * cin * cout * k * k
    for mode in ("exact", "split"):
        hip.set_conv_mode(mode)
        for _ in range(3):
            hip.gconv2d(x, wp, bp, cout, k, 1, k // 2, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            hip.gconv2d(x, wp, bp, cout, k, 1, k // 2, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("conv %s %-5s: %7.1f us  %6.1f TFLOP/s" % ((n, cin, cout, h, w, k), mode, dt * 1e6, flop / dt / 1e12))
