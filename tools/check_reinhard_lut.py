#!/usr/bin/env python3
"""Dev check (GPU box): table-driven vs exact Lab arithmetic of the float32 Reinhard path -- accuracy and kernel times.
usage: python tools/check_reinhard_lut.py [--oracle]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

import ct_hip

H, W = 1080, 1920
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def cases():
    rng = np.random.default_rng(5)
    u = rng.random((H, W, 3), dtype=np.float32)
    yield "uniform", u, rng.random((H, W, 3), dtype=np.float32)
    yield "u8", (rng.integers(0, 256, (H, W, 3)).astype(np.float32) / 255), (rng.integers(0, 256, (H, W, 3)).astype(np.float32) / 255)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    g = np.stack([xx / W, yy / H, (xx + yy) / (W + H)], -1).astype(np.float32)
    yield "graded", (0.8 * g + 0.1 * u).astype(np.float32), (0.5 * g[::-1] + 0.3).astype(np.float32)
    yield "dark/toe", (u * 0.12).astype(np.float32), (u[::-1] * 0.2).astype(np.float32)
    kink = u.copy()                                           # a third of the values sit within 4 ulp of the sRGB kink
    sel = rng.random((H, W, 3)) < 0.33
    kink[sel] = (np.float32(0.04045) + rng.integers(-4, 5, (H, W, 3)).astype(np.float32) * np.float32(2.0 ** -28))[sel]
    yield "kink", kink.astype(np.float32), u
    o = u.copy()
    o[::7, ::5] = 1.5
    o[::11, ::3] = -0.25
    yield "out-of-range", o, u


def run(t, r, mode):
    ct_hip.set_lab_mode(mode)
    T, R = torch.from_numpy(t).to(dev), torch.from_numpy(r).to(dev)
    st, sr = ct_hip.lab_stats(T), ct_hip.lab_stats(R)
    lab = ct_hip.reinhard_apply(T, st, sr, to_lab=True)
    out = ct_hip.reinhard(T, R)
    torch.cuda.synchronize()
    return st.cpu().numpy()[0], sr.cpu().numpy()[0], lab.cpu().numpy(), out.cpu().numpy()


def main():
    for name, t, r in (cases() if "--time-only" not in sys.argv else []):
        e = run(t, r, "exact")
        f = run(t, r, "table")
        print("%-13s stats |d| t %.2e r %.2e   Lab probe max|d| %.3e   RGB max|d| %.3e  (nan: %d/%d)" % (
            name, np.nanmax(np.abs(e[0][:6] - f[0][:6])), np.nanmax(np.abs(e[1][:6] - f[1][:6])),
            np.nanmax(np.abs(e[2].astype(np.float64) - f[2])), np.nanmax(np.abs(e[3].astype(np.float64) - f[3])),
            int(np.isnan(e[3]).sum()), int(np.isnan(f[3]).sum())), flush=True)
        if "--oracle" in sys.argv and name in ("uniform", "dark/toe"):
            from oracle import lab as olab, linear as olin
            ref = olin.color_transfer_between_images(t.astype(np.float64), r.astype(np.float64))
            for m, x in (("exact", e), ("table", f)):
                print("      vs oracle (%s): RGB %.3e  Lab-of-RGB %.3e" % (
                    m, np.abs(x[3] - ref).max(), np.abs(olab.rgb2lab(x[3].astype(np.float64)) - olab.rgb2lab(ref)).max()), flush=True)
    if "--time-only" in sys.argv:
        return timing()
    # constant target: inf / nan like the reference in both modes
    c = np.full((64, 64, 3), 0.3, np.float32)
    u = np.random.default_rng(1).random((64, 64, 3), dtype=np.float32)
    for mode in ("exact", "table"):
        o = run(c, u, mode)[3]
        print("constant target, %s: finite fraction %.3f" % (mode, np.isfinite(o).mean()))

    timing()


def timing():
    # timing, B = 4 pairs per call like bench.py
    B = 4
    rng = np.random.default_rng(0)
    T = torch.from_numpy(rng.random((B, H, W, 3), dtype=np.float32)).to(dev)
    R = torch.from_numpy(rng.random((B, H, W, 3), dtype=np.float32)).to(dev)
    out = torch.empty_like(T)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e_ in ev:
        e_.record()
    torch.cuda.synchronize()
    for mode in (("exact", "table") if "--time-only" not in sys.argv else ("table",)):
        ct_hip.set_lab_mode(mode)
        for _ in range(20):
            ct_hip.reinhard(T, R, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            ct_hip.reinhard(T, R, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ct_hip.profile_events(ev)
        ts, ta = [], []
        for _ in range(30):
            ct_hip.reinhard(T, R, out=out)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
            ta.append(ev[2].elapsed_time(ev[3]) * 1e3)
        ct_hip.profile_events(None)
        plane = H * W * 12
        print("%-6s %.1f us/step = %.0f pairs/s | stats %.1f us (%.2f TB/s, %.3f of 8) | apply %.1f us (%.2f TB/s, %.3f of 8)" % (
            mode, dt * 1e6, B / dt, np.mean(ts), 2 * B * plane / np.mean(ts) / 1e6, 2 * B * plane / np.mean(ts) / 8e6,
            np.mean(ta), 2 * B * plane / np.mean(ta) / 1e6, 2 * B * plane / np.mean(ta) / 8e6), flush=True)


if __name__ == "__main__":
    main()
