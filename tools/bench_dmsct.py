#!/usr/bin/env python3
"""Time DMSCT.forward (matcher + encoder x2 + fusion + decoder + head) at HxW (default 540x960), random init."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
from methods.dmsct import DMSCT

H = int(sys.argv[1]) if len(sys.argv) > 1 else 540
W = int(sys.argv[2]) if len(sys.argv) > 2 else 960
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
m = DMSCT().cuda().eval()
a, b = torch.rand(1, 3, H, W, device="cuda"), torch.rand(1, 3, H, W, device="cuda")
def t_ms(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = t_ms(lambda: m(a, b), n)
match = t_ms(lambda: m.match(a, b), n)
pad = m.derive_pad_size(a.shape)
ap = torch.nn.functional.pad(a, pad, mode="replicate")
both = torch.cat([ap, ap], dim=0)
enc = t_ms(lambda: m.encoder(both), n)                 # both views in one batch, as DMSCT.forward runs it
feats = m.encoder(ap)
fused = [torch.cat([f, f, f[:, :1]], 1).contiguous() for f in feats]
dec = t_ms(lambda: m.head(m.decoder(*fused)), n)
print("DMSCT %dx%d: forward %.2f ms (%.2f pairs/s) = matcher %.2f + encoder (both views, one batch) %.2f + decoder/head %.2f + fusion %.2f" % (
    H, W, full, 1e3 / full, match, enc, dec, full - match - enc - dec))
