#!/usr/bin/env python3
"""regrain at 1080p with k sweeps per launch (CT_HIP_REGRAIN_K, default 8; 1 = the round-2 one-launch-per-sweep path): time, and the
results saved under /tmp for a bitwise comparison between values of k (rg_sweepk_kernel recomputes a halo: same bits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "color-transfer_amd")); sys.path.insert(0, ROOT)
import torch, subprocess
import ct_hip
H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
t = torch.rand(H, W, 3, generator=g, dtype=torch.float64).cuda(); c = (t * 0.8 + 0.1 * torch.rand(H, W, 3, generator=g, dtype=torch.float64).cuda())
out = ct_hip.regrain(t, c)
torch.cuda.synchronize()
for _ in range(3): ct_hip.regrain(t, c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ct_hip.regrain(t, c)
torch.cuda.synchronize(); print("K=%s regrain 1080p: %.3f ms" % (os.environ.get("CT_HIP_REGRAIN_K", "8"), (time.perf_counter() - t0) / 10 * 1e3))
torch.save(out.cpu(), "/tmp/rg_%s.pt" % os.environ.get("CT_HIP_REGRAIN_K", "8"))
# odd sizes
for (h, w) in ((270, 481), (97, 133), (45, 61)):
    tt = torch.rand(h, w, 3, generator=g, dtype=torch.float64).cuda(); cc = torch.rand(h, w, 3, generator=g, dtype=torch.float64).cuda()
    torch.save(ct_hip.regrain(tt, cc).cpu(), "/tmp/rg_%s_%dx%d.pt" % (os.environ.get("CT_HIP_REGRAIN_K", "8"), h, w))
