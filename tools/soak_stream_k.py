#!/usr/bin/env python3
"""Soak of the shared (stream-K) form of the tile convolution: random shapes whose unit counts make the launcher choose it, each
run twice (bitwise repeatable) and against the whole-unit form (within the kernel's bound); the scratch's flag words stay zero and
no consumer ever gives up.  Usage: soak_stream_k.py [launch pairs]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(5)
torch.manual_seed(5)
shared = worst = 0
for it in range(n_iter):
    kh, kw = rng.choice(((3, 3), (1, 5), (5, 1), (1, 1)))
    cin, cout = 16 * rng.randint(1, 12), rng.choice((64, 96, 128, 192, 256))
    groups = (cout + 63) // 64
    while True:
        n, h, w = rng.randint(1, 3), rng.randint(40, 200), 4 * rng.randint(16, 80)
        units = n * ((h + 7) // 8) * ((w + 31) // 32) * groups
        r = units / 512.0
        if units > 512 and -(-units // 512) > 1.08 * r and units < 4000:
            break
    x = torch.randn(n, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, kh, kw, device="cuda") / (cin * kh * kw) ** 0.5
    wp, bp = ct_hip.pack_gconv_weight(wt, torch.randn(cout, device="cuda"))
    a = ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=rng.choice((0, 2, 3)))
    b = ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=0)
    b2 = ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=0)
    assert torch.equal(b, b2), ("not repeatable", kh, kw, cin, cout, n, h, w)
    ct_hip.set_conv_stream_k(False)
    c = ct_hip.gconv2d(x, wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2), act=0)
    ct_hip.set_conv_stream_k(True)
    d = (b - c).abs().max().item()
    worst = max(worst, d)
    shared += int(not torch.equal(b, c))
    assert d < 2e-4 and torch.isfinite(a).all(), (d, kh, kw, cin, cout, n, h, w)
torch.cuda.synchronize()
print("stream-K soak: %d shapes, %d took the shared form, worst |shared - whole| %.2e, scratch state %s" % (n_iter, shared, worst, ct_hip.conv_stream_k_state()))
assert ct_hip.conv_stream_k_state() == (0, 0)
