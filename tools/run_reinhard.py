#!/usr/bin/env python3
"""Profiling target: a few fused Reinhard calls (B pairs of 1080p float32) in the current lab mode.  usage: run_reinhard.py [n] [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(0)
T = torch.rand((B, 1080, 1920, 3), device="cuda", generator=g)
R = torch.rand((B, 1080, 1920, 3), device="cuda", generator=g)
out = torch.empty_like(T)
for _ in range(n):
    ct_hip.reinhard(T, R, out=out)
torch.cuda.synchronize()
print("done", ct_hip.lab_mode())
