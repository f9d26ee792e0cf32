#!/usr/bin/env python3
"""Times the fused Reinhard entries on 1080p pairs resident in HBM (HIP events, median of rounds).
env: CT_HIP_REINHARD_PERSIST=1 -> the fused float32 entries take the persistent launch (default: the two sweeps); CT_HIP_PERSIST_WGS=n.
usage: tools/bench_reinhard.py [pairs=16] [u8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import numpy as np, torch, ct_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
u8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
if u8:
    t, r, gt = (torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(3))
else:
    t, r, gt = (torch.rand((B, H, W, 3), device="cuda", generator=g) for _ in range(3))
out = torch.empty((B, H, W, 3), dtype=torch.float32, device="cuda")
psnr = torch.empty((B, 2), dtype=torch.float64, device="cuda")

def timed(fn, n=40, rounds=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e-3)
    return float(np.median(res)), float(np.min(res))

cfg = "persist=%s" % os.environ.get("CT_HIP_REINHARD_PERSIST", "0")
if u8:
    f_psnr = lambda: ct_hip.reinhard_persist(t, r, gt=gt, out=out, psnr_out=psnr)
    f_plain = lambda: ct_hip.reinhard_persist(t, r, out=out)
else:
    f_psnr = lambda: ct_hip.reinhard_psnr(t, r, gt, out=out, psnr_out=psnr)
    f_plain = lambda: ct_hip.reinhard(t, r, out=out)
for name, fn, planes in (("with PSNR", f_psnr, 4), ("transfer only", f_plain, 3)):
    med, mn = timed(fn)
    print("[%s %s B=%d] %-14s %8.1f us/call  %7.2f us/pair  %8.0f pairs/s (best %8.0f)  3-plane frac of 8 TB/s %.3f" % (
        cfg, "u8" if u8 else "f32", B, name, med * 1e6, med / B * 1e6, B / med, B / mn, 3 * H * W * 12 * B / med / 8e12), flush=True)
ws = ct_hip.workspace(ct_hip.CT_WS_REINHARD_PERSIST, H * W, B, t.device)
print("error word:", int(ws[:4].view(torch.int32)[0].item()), " psnr[0]:", psnr[0].tolist())
