#!/usr/bin/env python3
"""Time the streaming attention entry points at DCMCS3DI's 1080p shapes (rows of 1920 tokens, 64 channels, 96-channel values) and at
GMFlow's (windows of 448 tokens, 128 channels); CT_HIP_ATT16=0 selects the three-piece bf16 kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


lib = ct_hip.lib()
torch.manual_seed(0)
n, w = 1080, 1920
q, k = torch.randn(n, w, 64, device="cuda"), torch.randn(n, w, 64, device="cuda")
v = torch.randn(n, w, 96, device="cuda")
out, stats, colsum = torch.empty(n, w, 96, device="cuda"), torch.empty(n, w, 2, device="cuda"), torch.empty(n, w, device="cuda")
s = 1.0 / 64
t1 = timeit(lambda: ct_hip.check(lib.ct_attention_rows64_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None, n, w, s, None)))
t2 = timeit(lambda: ct_hip.check(lib.ct_attention_rows64_f32(q.data_ptr(), k.data_ptr(), None, None, stats.data_ptr(), n, w, s, None)))
t3 = timeit(lambda: ct_hip.check(lib.ct_attention_colsum64_f32(q.data_ptr(), k.data_ptr(), stats.data_ptr(), colsum.data_ptr(), n, w, s, None)))
print("ATT16=%s rows64 1080x1920: attend %.3f ms  stats %.3f ms  colsum %.3f ms" % (os.environ.get("CT_HIP_ATT16", "1"), t1, t2, t3))
b, l = 256, 448
q, k, v = (torch.randn(b, l, 128, device="cuda") for _ in range(3))
t4 = timeit(lambda: ct_hip.attention_tokens(q, k, v, None))
print("ATT16=%s tokens128 256 windows x 448: %.3f ms" % (os.environ.get("CT_HIP_ATT16", "1"), t4))
