#!/usr/bin/env python3
"""64->64 3x3 convolution (the DCMCS3DI ResB conv): conv_ws (weights stationary) vs conv_split (CT_HIP_CONV_WS=0), + max
difference between the two and against a float64 torch reference on a small crop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
import torch
import ct_hip as hip
n, cin, cout, h, w = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (2, 64, 64, 512, 512)
torch.manual_seed(0)
x = torch.randn(n, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda"); res = torch.randn(n, cout, h, w, device="cuda")
wp, bp = hip.pack_conv_weight(wt, b)
out = torch.empty(n, cout, h, w, device="cuda")
flop = 2.0 * n * h * w * cin * cout * 9
for r in (None, res):
    for _ in range(3):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=r, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=r, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("conv %s WS=%s residual=%s: %7.1f us  %6.1f TFLOP/s f32-equivalent (x6 = %.3f of 2.5 PF bf16)" % (
        (n, cin, cout, h, w), os.environ.get("CT_HIP_CONV_WS", "1"), r is not None, dt * 1e6, flop / dt / 1e12, 6 * flop / dt / 2.5e15))
ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x[:1, :, :64, :96].double(), wt.double(), b.double(), padding=1), 0.01) + res[:1, :, :64, :96].double()
print("max |hip - f64 torch| on the interior of a 64x96 crop: %.3e" % (out[:1, :, :63, :95].double() - ref[:, :, :63, :95]).abs().max().item())
