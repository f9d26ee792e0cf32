#!/usr/bin/env python3
"""Diagnostic: phase breakdown (s_memtime stamps) of conv_wino4_kernel on a 64->64 3x3 conv, batch 2 (default 1080p).
needs  SRC=conv_wino4 tools/build_variant.sh w4prof -DCT_W4_PROFILE ; run with CT_HIP_LIB=.../libct_tune_w4prof.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
os.environ.setdefault("CT_HIP_LIB", os.path.join(ROOT, "color-transfer_amd", "ct_hip", "libct_tune_w4prof.so"))
os.environ["CT_HIP_CONV_WINO"] = "1"
import numpy as np
import torch
import ct_hip
N, C = 2, 64
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
use_res = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
x = torch.randn(N, C, H, W, device="cuda")
wt = torch.randn(C, C, 3, 3, device="cuda") / 24
b = torch.randn(C, device="cuda")
wp, bp = ct_hip.pack_conv_weight(wt, b)
out = torch.empty_like(x)
prof = torch.zeros((256 * 4, 10), dtype=torch.int64, device="cuda")
ct_hip.lib().ct_conv_wino4_set_prof.argtypes = [ctypes.c_void_p]
ct_hip.lib().ct_conv_wino4_set_prof(ctypes.c_void_p(prof.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        e0.record()
    ct_hip.conv2d(x, wp, bp, C, 3, act=1, residual=x if use_res else None, out=out)
e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64).reshape(256, 4, 10)
names = ["head: maxima, touches, skip requests, fragments", "phase X: M A (blocks 0, 1), 48 MFMAs, T(s + 1)", "M A of blocks 2, 3 -> LDS", "barrier 1",
         "phase Y: 48 MFMAs of step s + 1, output side of s", "landed pair (wait, fix-ups, maximum) + barrier 2"]
print("kernel %.1f us (stamped build, skip=%s); s_memtime ticks per wave, median over workgroups" % (e0.elapsed_time(e1) * 1e3, use_res))
for w in (0, 3):
    q = p[:, w, :]
    tot = q.sum(axis=1)
    print("  wave %d" % w)
    for i, n in enumerate(names):
        print("    %-46s %10.0f  (%5.1f %%)" % (n, np.median(q[:, i]), 100 * np.median(q[:, i] / tot)))
    print("    total %.0f" % np.median(tot))
