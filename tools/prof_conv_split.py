#!/usr/bin/env python3
"""Diagnostic: phase breakdown (s_memtime stamps) of conv_split_kernel.  usage: prof_conv_split.py [res] [N C COUT H W KH KW F16]
(default: 64 -> 64 3x3, batch 2, 512x512, bf16 form; CT_HIP_CONV_WS=0 keeps 3x3 / cin <= 64 on the tile kernel)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
import numpy as np
import torch
import ct_hip
lib = ctypes.CDLL(os.path.join(ROOT, "color-transfer_amd", "ct_hip", "libct_conv_prof.so"))
P = ctypes.c_void_p
lib.ct_conv2d_split_prof_f32.argtypes = [P, P, P, P, P] + [ctypes.c_int] * 5 + [P] + [ctypes.c_int] * 4 + [P]
nums = [int(v) for v in sys.argv[1:] if v.lstrip("-").isdigit()]
N, C, CO, H, W, KH, KW, F16 = (nums + [2, 64, 64, 512, 512, 3, 3, 0][len(nums):])[:8]
x = torch.randn(N, C, H, W, device="cuda")
wt = torch.randn(CO, C, KH, KW, device="cuda") / (C * KH * KW) ** 0.5
b = torch.randn(CO, device="cuda")
ws, b64 = ct_hip.pack_conv_weight_split(wt, b)
w_exp = 0
if F16:
    ws, w_exp = ct_hip.pack_conv_weight_split16(wt)
out = torch.empty((N, CO, H, W), device="cuda")
prof = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
res = out.clone().data_ptr() if (len(sys.argv) > 1 and sys.argv[1] == "res") else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        e0.record()
    rc = lib.ct_conv2d_split_prof_f32(x.data_ptr(), ws.data_ptr(), b64.data_ptr(), res, out.data_ptr(), N, C, CO, H, W, prof.data_ptr(),
                                      F16, int(w_exp), KH, KW, None)
    assert rc == 0, rc
e1.record(); torch.cuda.synchronize()
pall = prof.cpu().numpy().astype(np.float64)
names = ["store_tile/acc-init", "barrier (tile visible)", "tap work (LDS reads + MFMA)", "epilogue", "barrier (tile free)", "per-tap barriers",
         "weights landed wait (wave 3)"]
print("kernel %.1f us (stamped build); s_memtime ticks per workgroup, median over workgroups" % (e0.elapsed_time(e1) * 1e3))
for label, p in (("wave 0", pall[:512]), ("wave 3 (weight streamer)", pall[512:])):
    p = p[p[:, :7].sum(axis=1) > 0]
    tot = p[:, :7].sum(axis=1)
    print(" ", label)
    for i, n in enumerate(names):
        print("    %-32s %10.0f  (%5.1f %%)" % (n, np.median(p[:, i]), 100 * np.median(p[:, i] / tot)))
    print("    total %.0f" % np.median(tot))
