#!/usr/bin/env python3
"""Stage timeline of the persistent Reinhard launch from the stamped diagnostic build (make ... EXTRA=-DCT_RP_STAMPS,
CT_HIP_LIB=.../libct_tune_stamps.so).  Stages: S(0) S(1) A(0) S(2) A(1) S(3) ...; per stage four s_memrealtime (100 MHz) values:
S: start | R done | T done | published;   A: start | statistics collected (wave 0) | barrier left | applied.
Prints mean / max lengths over the workgroups for the first and the last wave of a workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import numpy as np, torch, ct_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
t, r, gt = (torch.rand((B, H, W, 3), device="cuda", generator=g) for _ in range(3))
out = torch.empty_like(t)
if len(sys.argv) > 2 and sys.argv[2] == "u8":        # uint8 frames (the u8 front door)
    t, r, gt = ((x * 255).round().to(torch.uint8) for x in (t, r, gt))
for _ in range(20):
    ct_hip.reinhard_persist(t, r, gt=gt, out=out)
torch.cuda.synchronize()
ws = ct_hip.workspace(ct_hip.CT_WS_REINHARD_PERSIST, H * W, B, t.device)
need = ct_hip.lib().ct_workspace_bytes(ct_hip.CT_WS_REINHARD_PERSIST, H * W, B)
G, NS = 256, 2 + 2 * B
n_st = G * 2 * NS * 4
st = ws[need - n_st * 8:need].view(torch.int64).cpu().numpy().reshape(G, 2, NS, 4).astype(np.float64) / 100.0     # us
t0 = st[:, :, 0, 0].min()
end = st[:, :, NS - 1, 3].max()
print("launch spread: %.2f us; total %.1f us = %.2f us/pair" % (st[:, 0, 0, 0].max() - t0, end - t0, (end - t0) / B))
def f(x): return "%.2f/%.2f" % (x.mean(), x.max())
for stg in (3, 4, NS // 2 | 1, (NS // 2) & ~1, NS - 3, NS - 2):
    for wv, name in ((0, "wave 0 "), (1, "last wv")):
        s = st[:, wv, stg]
        if stg < 2 or stg & 1:
            print("stage %2d S(%2d) %s: R %s  T %s  publish %s   | start spread %.2f, T-done spread %.2f" % (stg, (stg + 1) // 2, name, f(s[:, 1] - s[:, 0]), f(s[:, 2] - s[:, 1]),
                  f(s[:, 3] - s[:, 2]), s[:, 0].max() - s[:, 0].min(), s[:, 2].max() - s[:, 2].min()))
        else:
            mid = s[:, 1] if wv == 0 else s[:, 0]
            print("stage %2d A(%2d) %s: collect/wait %s  (barrier left - start) %s  apply %s" % (stg, (stg - 2) // 2, name, f(mid - s[:, 0]), f(s[:, 2] - s[:, 0]), f(s[:, 3] - s[:, 2])))
