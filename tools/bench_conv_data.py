import os, sys, time
sys.path.insert(0, "color-transfer_amd")
import torch
import ct_hip as hip
n, cin, cout, h, w = 2, 64, 64, 1080, 1920
for name, gen in (("randn", lambda *s: torch.randn(*s, device="cuda")), ("zeros", lambda *s: torch.zeros(*s, device="cuda")),
                  ("small ints", lambda *s: torch.randint(-2, 3, s, device="cuda").float())):
    x = gen(n, cin, h, w); wt = gen(cout, cin, 3, 3) / 24
    b = torch.zeros(cout, device="cuda"); res = gen(n, cout, h, w)
    wp, bp = hip.pack_conv_weight(wt, b)
    out = torch.empty(n, cout, h, w, device="cuda")
    for _ in range(3):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=res, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hip.conv2d(x, wp, bp, cout, 3, act=1, residual=res, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-10s WS=%s: %7.1f us" % (name, os.environ.get("CT_HIP_CONV_WS", "1"), dt * 1e6))
