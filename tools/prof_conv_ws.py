#!/usr/bin/env python3
"""Diagnostic: phase breakdown (s_memtime stamps) of conv_ws_kernel on a 64->64 3x3 conv, batch 2 (default 1080p), through
ct_hip.conv2d (fp16 two-piece form by default; CT_HIP_CONV_WS16=0 = bf16 three-piece form).
needs `make -C color-transfer_amd/csrc prof`; run with CT_HIP_LIB=color-transfer_amd/ct_hip/libct_tune_wsprof.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
os.environ.setdefault("CT_HIP_LIB", os.path.join(ROOT, "color-transfer_amd", "ct_hip", "libct_tune_wsprof.so"))
import numpy as np
import torch
import ct_hip
N, C = 2, 64
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
x = torch.randn(N, C, H, W, device="cuda")
wt = torch.randn(C, C, 3, 3, device="cuda") / 24
b = torch.randn(C, device="cuda")
wp, bp = ct_hip.pack_conv_weight(wt, b)
out = torch.empty_like(x)
prof = torch.zeros((256 * 8, 8), dtype=torch.int64, device="cuda")
ct_hip.lib().ct_conv_ws_set_prof.argtypes = [ctypes.c_void_p]
ct_hip.lib().ct_conv_ws_set_prof(ctypes.c_void_p(prof.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        e0.record()
    ct_hip.conv2d(x, wp, bp, C, 3, act=1, residual=x, out=out)
e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64).reshape(256, 8, 8)
names = ["X: partial sums -> LDS", "barrier (one per step)", "Y: staging (scale, pieces -> LDS)", "Y: requests (skip row, input row)", "other (prologue, weights)",
         "Y: row maximum", "X: reduce r-1 + MFMAs"]
print("kernel %.1f us (stamped build, ws16=%s); s_memtime ticks per wave, median over workgroups" % (e0.elapsed_time(e1) * 1e3, ct_hip.conv_ws16()))
for w in (0, 4):
    q = p[:, w, :7]
    tot = q.sum(axis=1)
    print("  wave %d (mt = %d)" % (w, w // 4))
    for i, n in enumerate(names):
        print("    %-32s %10.0f  (%5.1f %%)" % (n, np.median(q[:, i]), 100 * np.median(q[:, i] / tot)))
    print("    total %.0f" % np.median(tot))
