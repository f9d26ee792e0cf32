#!/usr/bin/env python3
"""The persistent Reinhard launch on 16 resident 1080p pairs with the per-frame PSNR: float32 frames (ct_reinhard_persist_f32) and
uint8 frames (ct_reinhard_psnr_u8).  Prints pairs/s (wall clock over 100 calls, twice) and a checksum of the results, so that two
builds (CT_HIP_LIB=...) can be compared for speed AND for bitwise equal output.  Also the rocprofv3 target of the persist kernels."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import torch, ct_hip
B, H, W = 16, 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
t, r, gt = (torch.rand((B, H, W, 3), device="cuda", generator=g) for _ in range(3))
t[0, :64] *= 0.02                     # dark rows: the toe of both transforms
t8, r8, g8 = ((x * 255).round().to(torch.uint8) for x in (t, r, gt))
out = torch.empty_like(t)
ps = torch.empty((B, 2), dtype=torch.float64, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, args in (("float32", (t, r, gt)), ("uint8", (t8, r8, g8))):
    for rep in range(2):
        for _ in range(10):
            ct_hip.reinhard_persist(args[0], args[1], gt=args[2], out=out, psnr_out=ps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ct_hip.reinhard_persist(args[0], args[1], gt=args[2], out=out, psnr_out=ps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-8s %8.0f pairs/s" % (name, n * B / dt), flush=True)
    h = hashlib.sha1(out.cpu().numpy().tobytes() + ps.cpu().numpy().tobytes()).hexdigest()[:16]
    print("%-8s out+psnr sha1 %s  psnr[0] %r  status %d" % (name, h, ps[0].tolist(), ct_hip.device_status()))
