#!/usr/bin/env python3
"""64->64 3x3 convolution (the DCMCS3DI ResB conv) through ct_hip.conv2d: time + error against float64 torch on a crop, for the
fp16 two-piece weight-stationary kernel (default), the bf16 three-piece one (CT_HIP_CONV_WS16=0) and conv_split (CT_HIP_CONV_WS=0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
import torch
import ct_hip as hip
n, cin, cout, h, w = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (2, 64, 64, 512, 512)
xscale = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
torch.manual_seed(0)
x = torch.randn(n, cin, h, w, device="cuda") * xscale; wt = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
x[:, :, : h // 2] *= 1e-3                          # rows of very different magnitude inside one image
b = torch.randn(cout, device="cuda"); res = torch.randn(n, cout, h, w, device="cuda")
wp, bp = hip.pack_conv_weight(wt, b)
out = torch.empty(n, cout, h, w, device="cuda")
flop = 2.0 * n * h * w * cin * cout * 9
tag = "WS=%s WS16=%s WINO=%s" % (os.environ.get("CT_HIP_CONV_WS", "1"), os.environ.get("CT_HIP_CONV_WS16", "1"), os.environ.get("CT_HIP_CONV_WINO", "0"))
for r in (None, res):
    for _ in range(3):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=r, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=r, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("conv %s %s residual=%s: %7.1f us  %6.1f TFLOP/s f32-equivalent" % ((n, cin, cout, h, w), tag, r is not None, dt * 1e6, flop / dt / 1e12))
for name, (ya, yb) in (("small rows", (0, 48)), ("boundary", (h // 2 - 24, h // 2 + 24)), ("large rows", (h - 48, h))):
    xs = x[:1, :, ya:yb, :96].double()
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xs, wt.double(), b.double(), padding=1), 0.01) + res[:1, :, ya:yb, :96].double()
    d = (out[:1, :, ya + 1:yb - 1, :95].double() - ref[:, :, 1:-1, :95]).abs().max().item()
    print("  %-10s max |hip - f64 torch| %.3e  (max |conv out| %.3g)" % (name, d, (ref - res[:1, :, ya:yb, :96].double())[:, :, 1:-1, :95].abs().max().item()))
