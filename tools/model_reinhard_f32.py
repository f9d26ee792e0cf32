#!/usr/bin/env python3
"""Error budget of the float32 Reinhard arithmetic (csrc/ct_color_lut.h) from its numpy model in tools/gen_lab_tables.py --
every device rounding emulated -- against the float64 oracle, per input class: forward transform alone (L, a, b), the
transferred Lab image, the final RGB and Lab of the final RGB.  The GPU reproduces these numbers digit for digit
(tools/diag_reinhard_f32.py); this script is how the table shapes and the difference forms were chosen without a GPU.
usage: tools/model_reinhard_f32.py [H W]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import gen_lab_tables as G  # noqa: E402
from oracle import lab as olab  # noqa: E402

f32, f64 = np.float32, np.float64


def special_inputs(h, w):
    rng = np.random.default_rng(5)
    u = rng.random((h, w, 3), dtype=np.float32)
    yield "uniform", u, rng.random((h, w, 3), dtype=np.float32)
    yield "u8", (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255), (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], -1).astype(np.float32)
    yield "graded", (0.8 * g + 0.1 * u).astype(np.float32), (0.5 * g[::-1] + 0.3).astype(np.float32)
    yield "dark", (u * 0.12).astype(np.float32), (u[::-1] * 0.2).astype(np.float32)
    yield "scale-1.9", (u * 0.5 + 0.25).astype(np.float32), u
    yield "scale-3.3", (u * 0.28 + 0.36).astype(np.float32), u


def main():
    E, F, Gt = G.build_all()
    K = G.consts(F, Gt)
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
    for name, t, r in special_inputs(h, w):
        lt, lr = olab.rgb2lab(t.astype(f64)), olab.rgb2lab(r.astype(f64))
        mt, sdt = lt.reshape(-1, 3).mean(0), lt.reshape(-1, 3).std(0)
        mr, sdr = lr.reshape(-1, 3).mean(0), lr.reshape(-1, 3).std(0)
        lab_ref = (lt - mt) * (sdr / sdt) + mr
        rgb_ref = olab.lab2rgb(lab_ref)
        fy, dxy, dyz = G.forward_apply(E, F, t)
        st = np.stack(G.forward_stats(E, F, t), -1).reshape(-1, 3).astype(f64)
        sr = np.stack(G.forward_stats(E, F, r), -1).reshape(-1, 3).astype(f64)
        sc = sr.std(0) / st.std(0)
        off = sr.mean(0) - sc * st.mean(0)
        gy, dx, dz = G.fma32(fy, f32(sc[0]), f32(off[0])), G.fma32(dxy, f32(sc[1]), f32(off[1])), G.fma32(dyz, f32(sc[2]), f32(off[2]))
        lab_dev = np.stack([G.fma32(gy, f32(116), f32(-16)), f32(500) * dx, f32(200) * dz], -1).astype(f64)
        out = G.inverse(Gt, K, gy, dx, dz)
        e_fwd = np.abs(np.stack([116.0 * fy.astype(f64) - 16, 500.0 * dxy.astype(f64), 200.0 * dyz.astype(f64)], -1) - lt).reshape(-1, 3).max(0)
        print("  %-10s scale %.2f | forward L/a/b %.1e %.1e %.1e | Lab %.2e | RGB %.2e | Lab(RGB) %.2e | mean err %.1e" % (
            name, sc.max(), e_fwd[0], e_fwd[1], e_fwd[2], np.abs(lab_dev - lab_ref).max(), np.abs(out - rgb_ref).max(),
            np.abs(olab.rgb2lab(out.astype(f64)) - olab.rgb2lab(rgb_ref)).max(), np.abs(st.mean(0) * [116, 500, 200] - [16, 0, 0] - mt).max()))


if __name__ == "__main__":
    main()
