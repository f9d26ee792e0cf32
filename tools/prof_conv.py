#!/usr/bin/env python3
"""Diagnostic: phase breakdown (s_memtime stamps) of conv_mfma_kernel<3,2> on a 64->64 conv, batch 2, 512x512."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd"))
import numpy as np
import torch
import ct_hip
lib = ctypes.CDLL(os.path.join(ROOT, "color-transfer_amd", "ct_hip", "libct_conv_prof.so"))
P = ctypes.c_void_p
lib.ct_conv2d_prof_f32.argtypes = [P, P, P, P, P] + [ctypes.c_int] * 5 + [P, P]
N, C, H, W = 2, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 512
x = torch.randn(N, C, H, W, device="cuda")
wt = torch.randn(C, C, 3, 3, device="cuda") / 24
b = torch.randn(C, device="cuda")
wp, bp = ct_hip.pack_conv_weight(wt, b)
out = torch.empty_like(x)
prof = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = lib.ct_conv2d_prof_f32(x.data_ptr(), wp.data_ptr(), bp.data_ptr(), x.data_ptr(), out.data_ptr(), N, C, C, H, W,
                                prof.data_ptr(), None)
    assert rc == 0, rc
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.ct_conv2d_prof_f32(x.data_ptr(), wp.data_ptr(), bp.data_ptr(), x.data_ptr(), out.data_ptr(), N, C, C, H, W,
                       prof.data_ptr(), None)
e1.record(); torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "nores":
    e0.record()
    lib.ct_conv2d_prof_f32(x.data_ptr(), wp.data_ptr(), bp.data_ptr(), None, out.data_ptr(), N, C, C, H, W, prof.data_ptr(), None)
    e1.record(); torch.cuda.synchronize()
    print("NO RESIDUAL variant:")
p = prof.cpu().numpy().astype(np.float64)
p = p[p[:, :5].sum(axis=1) > 0]
names = ["store_tile/acc-init", "barrier (tile visible)", "tap loop (MFMA + prefetch issue)", "epilogue",
         "barrier (tile free)"]
tot = p[:, :5].sum(axis=1)
print("kernel %.1f us (stamped build); per-workgroup wave-0 cycles (s_memtime = 100 MHz ticks? see ratio), median over 512 WGs" % (e0.elapsed_time(e1) * 1e3))
for i, n in enumerate(names):
    print("  %-36s %10.0f  (%5.1f %%)" % (n, np.median(p[:, i]), 100 * np.median(p[:, i] / tot)))
print("  total %.0f" % np.median(tot))
