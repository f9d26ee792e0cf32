#!/usr/bin/env python3
"""Phase breakdown of linear_split_kernel from a -DCT_LS_PROFILE build (CT_HIP_LIB=.../libct_tune_lsprof.so): s_memtime
(100 MHz constant clock) totals of wave 0 of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
lib = ct_hip.lib()
lib.ct_debug_ls_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
NAMES = ["prologue", "barrier1", "lds+mfma+split", "barrier2", "W copy", "epilogue", "-"]
for T, k, n, act in ((114688, 128, 128, 0), (114688, 256, 1024, 6), (114688, 1024, 128, 0), (14336, 128, 128, 0)):
    xs = [torch.randn(T, k, device="cuda") for _ in range(4)]
    w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
    for i in range(3): ct_hip.linear_tokens(xs[i], w, b, act=act, mode="split")
    lib.ct_debug_ls_prof(None, 1)
    ct_hip.linear_tokens(xs[3], w, b, act=act, mode="split")
    buf = (ctypes.c_ulonglong * 8)()
    lib.ct_debug_ls_prof(buf, 1)
    wgs = buf[7]; tot = sum(buf[:7])
    print("T=%d K=%d N=%d: %d workgroups, %.1f k ticks per workgroup (memtime 100 MHz)" % (T, k, n, wgs, tot / wgs / 100.0))
    print("   " + "  ".join("%s %.0f%%" % (NAMES[i], 100.0 * buf[i] / tot) for i in range(6)))
