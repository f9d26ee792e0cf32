#!/usr/bin/env python3
"""FFN1 (T = 114688, 256 -> 1024, GELU) on operands of different content: the matrix pipe's power draw depends on the data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
def t_ms(fn, n=24):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
T = 114688
w = torch.randn(1024, 256, device="cuda") / 16
for name, gen in (("randn", lambda: torch.randn(T, 256, device="cuda")), ("zeros", lambda: torch.zeros(T, 256, device="cuda")),
                  ("ones", lambda: torch.ones(T, 256, device="cuda")), ("small ints", lambda: torch.randint(-3, 4, (T, 256), device="cuda").float())):
    xs = [gen() for _ in range(6)]
    it = [0]
    def call():
        it[0] += 1
        return ct_hip.linear_tokens(xs[it[0] % 6], w, None, act=6, mode="split", partials=True)
    print("FFN1 x = %-10s %.3f ms" % (name, t_ms(call)))
wz = torch.zeros(1024, 256, device="cuda")
xs = [torch.zeros(T, 256, device="cuda") for _ in range(6)]
it = [0]
def call2():
    it[0] += 1
    return ct_hip.linear_tokens(xs[it[0] % 6], wz, None, act=6, mode="split", partials=True)
print("FFN1 x = zeros, w = zeros: %.3f ms" % t_ms(call2))
