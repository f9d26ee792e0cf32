#!/usr/bin/env python3
"""Round-5 diagnostic of the float32 Reinhard path on the GPU: max-abs errors against the float64 oracle (Lab before lab2rgb,
RGB, Lab of the RGB) for the branch-walking inputs of tests/test_linear_gpu.py, for the two sweeps, the persistent launch and
the uint8 front door, plus the statistics' error.  Prints one line per case; used to set the tolerances of the GPU tests.
usage: tools/diag_reinhard_f32.py [H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "color-transfer_amd")]
import numpy as np
import torch

import ct_hip
from oracle import lab as olab

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)


def special_inputs(h, w):
    rng = np.random.default_rng(5)
    u = rng.random((h, w, 3), dtype=np.float32)
    yield "uniform", u, rng.random((h, w, 3), dtype=np.float32)
    yield "u8", (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255), (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], -1).astype(np.float32)
    yield "graded", (0.8 * g + 0.1 * u).astype(np.float32), (0.5 * g[::-1] + 0.3).astype(np.float32)
    yield "dark", (u * 0.12).astype(np.float32), (u[::-1] * 0.2).astype(np.float32)
    kink = u.copy()
    sel = rng.random((h, w, 3)) < 0.33
    kink[sel] = (np.float32(0.04045) + rng.integers(-4, 5, (h, w, 3)).astype(np.float32) * np.float32(2.0 ** -28))[sel]
    yield "kink", kink, u
    o = u.copy()
    o[::7, ::5] = 1.5
    o[::11, ::3] = -0.25
    yield "out-of-range", o, u
    yield "wide1.9", (u * 0.5 + 0.25).astype(np.float32), u
    yield "wide3.4", (u * 0.28 + 0.36).astype(np.float32), u
    yield "wide6", (u * 0.15 + 0.4).astype(np.float32), u


def oracle(t, r):
    lt, lr = olab.rgb2lab(t.astype(np.float64)), olab.rgb2lab(r.astype(np.float64))
    mt, sdt = lt.reshape(-1, 3).mean(0), lt.reshape(-1, 3).std(0)
    mr, sdr = lr.reshape(-1, 3).mean(0), lr.reshape(-1, 3).std(0)
    lab = (lt - mt) * (sdr / sdt) + mr
    return lab, olab.lab2rgb(lab), (mt, sdt, mr, sdr)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def lab_of(rgb):
    return olab.rgb2lab(np.asarray(rgb, np.float64))


for mode in ("table", "exact"):
    ct_hip.set_lab_mode(mode)
    print("== mode %s, %dx%d" % (mode, H, W), flush=True)
    for name, t, r in special_inputs(H, W):
        lab_ref, rgb_ref, (mt, sdt, mr, sdr) = oracle(t, r)
        td, rd = dev(t), dev(r)
        st, sr = ct_hip.lab_stats(td), ct_hip.lab_stats(rd)
        s = st.cpu().numpy()[0]
        e_stat = max(np.abs(s[0:3] - mt).max(), np.abs(s[3:6] - sdt).max())
        probe = ct_hip.reinhard_apply(td, st, sr, to_lab=True).cpu().numpy().astype(np.float64)
        out = ct_hip.reinhard(td, rd).cpu().numpy()
        lab_rgb_ref = lab_of(rgb_ref)
        line = "  %-13s scale %.2f | stats %.1e | two-sweep: lab %.2e rgb %.2e lab(rgb) %.2e" % (
            name, (sdr / sdt).max(), e_stat, np.abs(probe - lab_ref).max(), np.abs(out - rgb_ref).max(), np.abs(lab_of(out) - lab_rgb_ref).max())
        if mode == "table" and ct_hip.reinhard_persist_supported(H * W):
            op = ct_hip.reinhard_persist(td[None], rd[None], verify=True)[0].cpu().numpy()
            line += " | persist: rgb %.2e lab(rgb) %.2e vs two-sweep %.2e" % (np.abs(op - rgb_ref).max(), np.abs(lab_of(op) - lab_rgb_ref).max(), np.abs(op - out).max())
        print(line, flush=True)
    if mode == "table":
        # the uint8 front door against the oracle on k / 255
        rng = np.random.default_rng(9)
        t8 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        r8 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        tf, rf = t8.astype(np.float32) / np.float32(255), r8.astype(np.float32) / np.float32(255)
        lab_ref, rgb_ref, _ = oracle(tf, rf)
        o8 = ct_hip.reinhard_persist(dev(t8)[None], dev(r8)[None], verify=True)[0].cpu().numpy()
        of = ct_hip.reinhard(dev(tf), dev(rf)).cpu().numpy()
        print("  u8 front door: rgb %.2e lab(rgb) %.2e | vs float32 two-sweep on k/255: %.2e" % (
            np.abs(o8 - rgb_ref).max(), np.abs(lab_of(o8) - lab_of(rgb_ref)).max(), np.abs(o8 - of).max()), flush=True)
ct_hip.set_lab_mode("table")
