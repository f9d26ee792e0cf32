#!/usr/bin/env python3
"""Times the two Reinhard sweeps separately on 1080p float32 pairs resident in HBM (HIP events): the statistics sweep alone
(ct_lab_stats_f32 over 2 B images: the same kernel as the fused call + a 5 us finalize), the fused call with and without the
per-frame PSNR; apply = fused - statistics.  CT_HIP_LIB selects a tuning build.  usage: tools/bench_sweeps.py [pairs=16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import numpy as np, torch, ct_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
both = torch.rand((2 * B, H, W, 3), device="cuda", generator=g)
t, r = both[:B], both[B:]
gt = torch.rand((B, H, W, 3), device="cuda", generator=g)
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
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(res))

ts = timed(lambda: ct_hip.lab_stats(both))
tp = timed(lambda: ct_hip.reinhard_psnr(t, r, gt, out=out, psnr_out=psnr))
tn = timed(lambda: ct_hip.reinhard(t, r, out=out))
MB = H * W * 12 * B / 1e6
print("[%s] B=%d  stats %.1f us (%.2f TB/s on 2 planes) | fused+PSNR %.1f us -> apply+PSNR %.1f (%.2f TB/s on 3 planes) | fused %.1f -> apply %.1f (%.2f TB/s on 2 planes) | %.0f / %.0f pairs/s" % (
    os.path.basename(os.environ.get("CT_HIP_LIB", "libct_hip.so")), B, ts, 2 * MB / ts, tp, tp - ts, 3 * MB / (tp - ts), tn, tn - ts, 2 * MB / (tn - ts), B / tp * 1e6, B / tn * 1e6))
