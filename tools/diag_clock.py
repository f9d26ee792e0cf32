#!/usr/bin/env python3
"""GPU box, diagnostic build (-DCT_DIAG_CLOCK): the shader clock the chip holds inside lab_moments_lut_kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, ct_hip
torch.cuda.set_device(0)
B = 4
g = torch.Generator(device="cuda").manual_seed(0)
T = torch.rand((B, 1080, 1920, 3), device="cuda", generator=g); R = torch.rand((B, 1080, 1920, 3), device="cuda", generator=g)
out = torch.empty_like(T)
ws = ct_hip.workspace(ct_hip.CT_WS_REINHARD, 1080 * 1920, B, T.device)
for burst in (1, 20, 200, 2000):
    for _ in range(burst):
        ct_hip.reinhard(T, R, out=out)
    torch.cuda.synchronize()
    w = ws.view(torch.float64)
    piv = w[2 * B * 1024 * 12: 2 * B * 1024 * 12 + 2 * B * 4].cpu().numpy().reshape(2 * B, 4)
    print("after a burst of %4d calls: in-kernel clock (GHz) per image: %s" % (burst, np.array2string(piv[:, 3], precision=3)))
