#!/usr/bin/env python3
"""Diagnostic: phase breakdown (s_memtime stamps) of conv_wino_kernel on a 64->64 3x3 conv, batch 2 (default 1080p).
needs  SRC=conv_wino tools/build_variant.sh wnprof -DCT_WN_PROFILE ; run with CT_HIP_LIB=.../libct_tune_wnprof.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
os.environ.setdefault("CT_HIP_LIB", os.path.join(ROOT, "color-transfer_amd", "ct_hip", "libct_tune_wnprof.so"))
os.environ["CT_HIP_CONV_WINO"] = "1"
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
ct_hip.lib().ct_conv_wino_set_prof.argtypes = [ctypes.c_void_p]
ct_hip.lib().ct_conv_wino_set_prof(ctypes.c_void_p(prof.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        e0.record()
    ct_hip.conv2d(x, wp, bp, C, 3, act=1, residual=x, out=out)
e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64).reshape(256, 8, 8)
names = ["requests + scale", "T: transform, pieces -> B image", "barrier after T", "fragment reads + barrier", "MFMAs + M image", "barrier after C",
         "D: output transform, epilogue, stores", "staging of the next rows + 2 barriers"]
print("kernel %.1f us (stamped build); s_memtime ticks per wave, median over workgroups" % (e0.elapsed_time(e1) * 1e3))
for w in (0, 7):
    q = p[:, w, :]
    tot = q.sum(axis=1)
    print("  wave %d" % w)
    for i, n in enumerate(names):
        print("    %-40s %10.0f  (%5.1f %%)" % (n, np.median(q[:, i]), 100 * np.median(q[:, i] / tot)))
    print("    total %.0f" % np.median(tot))
