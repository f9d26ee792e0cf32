#!/usr/bin/env python3
"""Profile target: the persistent Reinhard launch on 16 resident 1080p pairs with the per-frame PSNR -- 30 calls with float32
frames (ct_reinhard_persist_f32) and 30 with uint8 frames (ct_reinhard_psnr_u8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import torch, ct_hip
B, H, W = 16, 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
t, r, gt = (torch.rand((B, H, W, 3), device="cuda", generator=g) for _ in range(3))
t8, r8, g8 = ((x * 255).round().to(torch.uint8) for x in (t, r, gt))
out = torch.empty_like(t)
ps = torch.empty((B, 2), dtype=torch.float64, device="cuda")
for _ in range(30):
    ct_hip.reinhard_persist(t, r, gt=gt, out=out, psnr_out=ps)
torch.cuda.synchronize()
for _ in range(30):
    ct_hip.reinhard_persist(t8, r8, gt=g8, out=out, psnr_out=ps)
torch.cuda.synchronize()
print("ok", ps[0].tolist())
