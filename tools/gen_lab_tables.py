#!/usr/bin/env python3
"""Generates color-transfer_amd/csrc/ct_lab_tables.h: the three look-up tables of the float32 sRGB <-> Lab path
(csrc/ct_color_lut.h, round 5) and the constants that go with them, and carries the numpy MODEL of that arithmetic
(every device rounding emulated) that tests/test_lab_tables.py holds against the float64 oracle.

The functions tabulated are the scalar pieces of scikit-image 0.18.3's rgb2lab / lab2rgb as the reference calls them
(methods/linear.py:25,26,40).  All three tables have 16-byte entries {a0, a1, a2, node}: the value is
a0 + d (a1 + d a2) with d = x - node, ONE ds_read_b128 and three float32 instructions.

  E  sRGB gamma expansion  c -> c/12.92 (c <= 0.04045) | ((c+0.055)/1.055)**2.4,  c in [0,1].  Uniform grid; the entry
     address is (bits(fmaf(c, S, 1.5 * 2^19 + 0.5)) & 0x3fff0): S puts a cell boundary exactly between the two float32
     neighbours of 0.04045.
  F  Lab's f() on the SHIFTED argument u = v + c0:  v -> cbrt(v) (v > 0.008856) | 7.787 v + 16/116.  Log grid, 128 cells per
     octave, address (bits(u) >> 12) & 0x3ff0.  The shift c0 (~2^-7, folded into the first fma of the matrix row) keeps u in
     8 octaves down to v = 0, so the linear toe of f() is part of the table (no toe test, no clamp); c0 puts the 0.008856 kink
     on a cell boundary.  a0 is a multiple of 2^-24 below 1: DIFFERENCES of two a0 are exact in float32, so
     fx - fy = (a0x - a0y) + (rx - ry) carries no rounding of the big parts (a* = 500 (fx - fy) is what the 1e-4 gate binds).
  G  sRGB gamma compression on w = u + c1:  u -> 12.92 u (u <= 0.0031308) | 1.055 u**(1/2.4) - 0.055, for w clamped to
     [c1, 1 + c1].  Same log grid; c1 (~2^-8) puts the kink on a cell boundary; the end cells return exactly 0 and 1.

Nodes are NOT at the cell centres: each is the float32 near the centre for which a0 falls on its storage grid (float32, or
2^-24 for F), so that a0 is stored without a rounding error; (a1, a2) are least-squares fits on Chebyshev points of the cell.
Everything is verified here against mpmath (40 digits) on dense samples through the numpy emulation of the device arithmetic.
`--check` re-generates and compares with the committed header (tests/test_lab_tables.py).
"""
import os
import struct
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "color-transfer_amd", "csrc", "ct_lab_tables.h")

f32 = np.float32
f64 = np.float64

KINK_E = 0.04045          # skimage rgb2xyz (colorconv.py l.657)
KINK_F = 0.008856         # skimage xyz2lab (colorconv.py l.955)
KINK_FI = 0.2068966       # skimage lab2xyz (colorconv.py l.1024)
KINK_G = 0.0031308        # skimage xyz2rgb (colorconv.py l.615)

E_MAGIC = f32(1.5 * 2 ** 19 + 0.5)      # float32 ulp there is 2^-4: bits = bits(1.5 * 2^19) + round(16 (c S + 0.5))
E_MASK = 0x3FFF0
F_BITS = 7
F_KCELL = 9               # u_kink = 2^-6 (1 + 9/128)
G_BITS = 7
G_KCELL = 103             # w_kink = 2^-8 (1 + 103/128)

# skimage colorconv.py l.338-340 and the D65 / 2 degree white point (l.426); the same numbers as csrc/ct_color.h
XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], f64)
WHITE = np.array([0.95047, 1.0, 1.08883], f64)
M = XYZ_FROM_RGB / WHITE[:, None]                    # white-normalised rows
MI = np.linalg.inv(XYZ_FROM_RGB) * WHITE[None, :]    # columns scaled back
# accumulation order of a row: smallest weight first, so that the early roundings happen on small partial sums
ROW_ORDER = ((2, 1, 0), (2, 0, 1), (0, 1, 2))


def bits_of(x):
    return np.asarray(x, f32).view(np.uint32)


def fma32(a, b, c):
    """float32 fma: the product of two float32 is exact in float64; the double rounding of the sum is ~2^-29 rare"""
    return (np.asarray(a, f64) * np.asarray(b, f64) + np.asarray(c, f64)).astype(f32)


def lin_exact(c):
    c = np.asarray(c, f64)
    return np.where(c > KINK_E, ((np.maximum(c, 0) + 0.055) / 1.055) ** 2.4, c / 12.92)


def lab_f_exact(v):
    v = np.asarray(v, f64)
    return np.where(v > KINK_F, np.cbrt(np.maximum(v, 1e-300)), 7.787 * v + 16.0 / 116.0)


def compress_exact(u):
    u = np.asarray(u, f64)
    return np.where(u > KINK_G, 1.055 * np.maximum(u, 1e-300) ** (1 / 2.4) - 0.055, 12.92 * u)


def f32_grid(v):
    v = np.asarray(v, f64)
    return np.spacing(np.abs(v).astype(f32)).astype(f64) + (v == 0)


def fit_cell(fn, lo, hi, grid_of, spread=0.1, node=None, a0=None):
    """float32 node near the centre of [lo, hi] + quadratic a0 + d (a1 + d a2), d = x - node, with a0 ON grid_of(a0)"""
    c, h = (lo + hi) / 2, (hi - lo) / 2
    x = c + h * np.cos(np.pi * (np.arange(24) + 0.5) / 24)
    y = fn(x)
    if node is None:
        q = np.polyfit((x - c) / h, y, 2)
        sp = f64(np.spacing(f32(c)))
        m = int(2 * spread * h / sp) + 1
        step = max(1, m // 4000)
        cand = np.unique((c - spread * h + np.arange(0, m, step) * sp).astype(f32))
        cand = cand[(cand >= f32(lo)) & (cand <= f32(hi))]
        if len(cand) == 0:
            cand = np.array([f32(c)])
        v = np.polyval(q, (cand.astype(f64) - c) / h)
        g = grid_of(v)
        j = int(np.argmin(np.abs(v / g - np.rint(v / g))))
        node = cand[j]
        a0 = np.rint(v[j] / g[j]) * g[j]
    d = x - f64(node)
    sol = np.linalg.lstsq(np.stack([d, d * d], 1), y - a0, rcond=None)[0]
    return f32(node), f64(a0), f64(sol[0]), f64(sol[1])


# ---------------------------------------------------------------------------------------------------------------------
# table E
# ---------------------------------------------------------------------------------------------------------------------
def e_index_exact(c, S):
    """the entry index the device computes for the float32 c, in exact rational arithmetic (round to nearest even on the 2^-4 grid)"""
    t = Fraction(float(c)) * Fraction(float(S)) * 16 + 8          # (c S + 0.5) * 16; the 1.5 * 2^19 is a multiple of the grid
    n = t.numerator // t.denominator
    r = t - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (n & 1)):
        n += 1
    return n >> 4


def choose_scale():
    c_lo = f32(KINK_E)
    if float(c_lo) > KINK_E:
        c_lo = np.nextafter(c_lo, f32(0))
    c_hi = np.nextafter(c_lo, f32(1))
    assert float(c_lo) <= KINK_E < float(c_hi)
    k = 40
    s0 = f32((k + 0.46875) / KINK_E)
    for step in range(0, 64):
        for s in (s0 + f32(step) * np.spacing(s0), s0 - f32(step) * np.spacing(s0)):
            if e_index_exact(c_lo, s) == k and e_index_exact(c_hi, s) == k + 1:
                return f32(s), k
    raise RuntimeError("no float32 scale puts a cell boundary on the sRGB kink")


def build_E():
    S, k = choose_scale()
    n = e_index_exact(f32(1.0), S) + 1
    tab = np.zeros((n, 4), f32)
    for i in range(n):
        lo = max((i - 0.53125) / f64(S), 0.0)
        hi = min((i + 0.46875) / f64(S), 1.0)
        if i == 0:
            tab[i] = [0.0, f32(1 / 12.92), 0.0, 0.0]                 # lin(0) = 0 exactly
            continue
        if i <= k:                                                    # the linear segment c / 12.92
            node, a0, _, _ = fit_cell(lin_exact, lo, hi, f32_grid)
            tab[i] = [f32(a0), f32(1 / 12.92), 0.0, node]
            continue
        fn = lambda c: ((c + 0.055) / 1.055) ** 2.4                   # the power branch on the whole cell
        node, a0, a1, a2 = fit_cell(fn, lo, hi, f32_grid)
        tab[i] = [f32(a0), f32(a1), f32(a2), node]
        assert f64(tab[i, 0]) == a0
    return dict(S=S, k=k, n=n, tab=tab)


def e_index(E, c):
    y = fma32(c, E["S"], E_MAGIC)
    return ((bits_of(y) & E_MASK) >> 4).astype(np.int64)


def expand(E, c):
    c = np.asarray(c, f32)
    e = E["tab"][e_index(E, c)]
    d = (c - e[..., 3]).astype(f32)
    return fma32(d, fma32(d, e[..., 2], e[..., 1]), e[..., 0])


# ---------------------------------------------------------------------------------------------------------------------
# table F
# ---------------------------------------------------------------------------------------------------------------------
def build_F():
    ukink = 2.0 ** -6 * (1 + F_KCELL / 128.0)
    c0 = f32(ukink - KINK_F)
    assert float(c0) >= 2.0 ** -7
    top = f64(f32(1.0)) + f64(c0) * (1 + 1e-6)                                      # u never exceeds 1 + c0
    n = int(((bits_of(f32(top)) >> (23 - F_BITS)) & ((8 << F_BITS) - 1))) + 1
    kcell = (1 << F_BITS) + F_KCELL                                                 # first cell of the cube-root branch
    tab = np.zeros((n, 4), f32)
    grid24 = lambda v: np.full_like(np.asarray(v, f64), 2.0 ** -24)
    for i in range(n):
        o, m = i >> F_BITS, i & ((1 << F_BITS) - 1)
        lo = max(2.0 ** (o - 7) * (1 + m / 128.0), f64(c0))
        hi = min(2.0 ** (o - 7) * (1 + (m + 1) / 128.0), top)
        if i < kcell:
            fn = lambda u: 7.787 * (u - f64(c0)) + 16.0 / 116.0
            node, a0, _, _ = fit_cell(fn, lo, hi, grid24)
            tab[i] = [f32(a0), f32(7.787), 0.0, node]
        else:
            fn = lambda u: np.cbrt(u - f64(c0))
            node, a0, a1, a2 = fit_cell(fn, lo, hi, grid24)
            tab[i] = [f32(a0), f32(a1), f32(a2), node]
        assert f64(tab[i, 0]) == a0 and a0 < 1.0 and a0 * 2 ** 24 == np.rint(a0 * 2 ** 24)
    return dict(c0=c0, ukink=f32(ukink), n=n, kcell=kcell, tab=tab)


def f_index(u):
    return ((bits_of(u) >> (23 - F_BITS)) & ((8 << F_BITS) - 1)).astype(np.int64)


def lab_f_parts(F, u, ul=None):
    """(a0, r): f(u - c0) = a0 + r; ul = low part of a two-term argument (added to d)"""
    u = np.asarray(u, f32)
    e = F["tab"][f_index(u)]
    d = (u - e[..., 3]).astype(f32)
    if ul is not None:
        d = (d + ul).astype(f32)
    r = (d * fma32(d, e[..., 2], e[..., 1])).astype(f32)
    return e[..., 0], r


def lab_f_value(F, u):
    u = np.asarray(u, f32)
    e = F["tab"][f_index(u)]
    d = (u - e[..., 3]).astype(f32)
    return fma32(d, fma32(d, e[..., 2], e[..., 1]), e[..., 0])


# ---------------------------------------------------------------------------------------------------------------------
# table G
# ---------------------------------------------------------------------------------------------------------------------
def build_G():
    wkink = 2.0 ** -8 * (1 + G_KCELL / 128.0)
    c1 = f32(wkink - KINK_G)
    assert float(c1) >= 2.0 ** -8
    whi = f32(f32(1.0) + c1)
    first = 7 << G_BITS                                             # (exponent & 15) of 2^-8 is 7
    n = int(((bits_of(whi) >> (23 - G_BITS)) & ((16 << G_BITS) - 1))) - first + 1
    tab = np.zeros((n, 4), f32)
    for i in range(n):
        o, m = i >> G_BITS, i & ((1 << G_BITS) - 1)
        lo = max(2.0 ** (o - 8) * (1 + m / 128.0), f64(c1))
        hi = min(2.0 ** (o - 8) * (1 + (m + 1) / 128.0), f64(whi))
        if i == 0:
            tab[i] = [0.0, f32(12.92), 0.0, c1]                      # w = c1 (u = 0) -> exactly 0
        elif i < G_KCELL:
            fn = lambda w: 12.92 * (w - f64(c1))
            node, a0, _, _ = fit_cell(fn, lo, hi, f32_grid)
            tab[i] = [f32(a0), f32(12.92), 0.0, node]
        elif i == n - 1:                                             # w in [1, 1 + c1]: node on the clamp, exactly 1 there
            fn = lambda w: 1.055 * (w - f64(c1)) ** (1 / 2.4) - 0.055
            node, a0, a1, a2 = fit_cell(fn, lo, hi, f32_grid, node=whi, a0=1.0)
            tab[i] = [1.0, f32(a1), f32(a2), whi]
        else:
            fn = lambda w: 1.055 * (w - f64(c1)) ** (1 / 2.4) - 0.055
            node, a0, a1, a2 = fit_cell(fn, lo, hi, f32_grid)
            tab[i] = [f32(a0), f32(a1), f32(a2), node]
    return dict(c1=c1, whi=whi, n=n, first=first, tab=tab)


def compress(G, w):
    w = np.minimum(np.maximum(np.asarray(w, f32), G["c1"]), G["whi"])               # v_med3_f32
    idx = (((bits_of(w) >> (23 - G_BITS)) & ((16 << G_BITS) - 1)).astype(np.int64) - G["first"])
    e = G["tab"][idx]
    d = (w - e[..., 3]).astype(f32)
    return fma32(d, fma32(d, e[..., 2], e[..., 1]), e[..., 0])


# ---------------------------------------------------------------------------------------------------------------------
# the model of the device pipeline (ct_color_lut.h); returns what the kernels return, rounding for rounding
# ---------------------------------------------------------------------------------------------------------------------
def consts(F, G):
    rho = MI.sum(1)
    return dict(
        fwd=np.array([[f32(M[k, i]) for i in range(3)] for k in range(3)], f32),
        inv_r=np.array([rho[0], MI[0, 0], -MI[0, 2]], f64).astype(f32),             # r = rho y + i00 (x - y) - i02 (y - z)
        inv_g=np.array([rho[1], MI[1, 0], -MI[1, 2]], f64).astype(f32),
        inv_b=np.array([rho[2], MI[2, 0], MI[2, 0] + MI[2, 1]], f64).astype(f32),   # b = rho z + i20 (x - y) + (i20 + i21) (y - z)
        toe_inv=f32(KINK_FI), toe_a=f32(1 / 7.787), toe_b=f32(-(16.0 / 116.0) / 7.787))


def row_u(F, l, k, two):
    o = ROW_ORDER[k]
    p = fma32(l[o[1]], f32(M[k, o[1]]), fma32(l[o[0]], f32(M[k, o[0]]), F["c0"]))
    u = fma32(l[o[2]], f32(M[k, o[2]]), p)
    ul = fma32(l[o[2]], f32(M[k, o[2]]), (p - u).astype(f32)) if two else None      # the rounding error of the last fma (exact when p - u is)
    return u, ul


def forward_stats(E, F, rgb):
    """(fy, fx - fy, fy - fz) as the statistics sweep computes them: plain float32 values of f(), unbiased roundings"""
    l = [expand(E, rgb[..., i]) for i in range(3)]
    f = [lab_f_value(F, row_u(F, l, k, False)[0]) for k in range(3)]
    return f[1], (f[0] - f[1]).astype(f32), (f[1] - f[2]).astype(f32)


def forward_apply(E, F, rgb):
    """the apply sweep's forward transform: two-term arguments for X and Y, differences of the grid parts exact"""
    l = [expand(E, rgb[..., i]) for i in range(3)]
    parts = []
    for k in range(3):
        u, ul = row_u(F, l, k, k < 2)
        parts.append(lab_f_parts(F, u, ul))
    (cx, rx), (cy, ry), (cz, rz) = parts
    fy = (cy + ry).astype(f32)
    dxy = ((cx - cy).astype(f32) + (rx - ry).astype(f32)).astype(f32)
    dyz = ((cy - cz).astype(f32) + (ry - rz).astype(f32)).astype(f32)
    return fy, dxy, dyz


def inverse(G, K, gy, dx, dz):
    """(gy, gx - gy, gy - gz) -> clipped float32 sRGB: cubes in difference form, blue from z"""
    gy, dx, dz = (np.asarray(a, f32) for a in (gy, dx, dz))
    gy2 = (gy * gy).astype(f32)
    y = (gy2 * gy).astype(f32)
    t3 = (f32(3) * gy2).astype(f32)
    ux = (dx * fma32(dx, fma32(f32(3), gy, dx), t3)).astype(f32)                     # x - y = dx (3 gy^2 + dx (3 gy + dx))
    uz = (dz * fma32(-dz, fma32(f32(3), gy, -dz), t3)).astype(f32)                   # y - z
    gx, gz = (gy + dx).astype(f32), (gy - dz).astype(f32)
    z = ((gz * gz).astype(f32) * gz).astype(f32)
    thr = K["toe_inv"]
    toe = np.minimum(np.minimum(gx, gy), gz) <= thr
    if toe.any():
        # the toe block of ct_color_lut.h: gx^3 = y + ux and gz^3 = y - uz are already there; linear-branch components are tiny
        gzc = np.maximum(gz, f32(0))                                                  # lab2xyz: z < 0 -> 0
        xl, yl, zl = fma32(gx, K["toe_a"], K["toe_b"]), fma32(gy, K["toe_a"], K["toe_b"]), fma32(gzc, K["toe_a"], K["toe_b"])
        bx, by, bz = gx > thr, gy > thr, gz > thr
        yn = np.where(by, y, yl)
        dyc = (y - yn).astype(f32)
        ux = np.where(toe, np.where(bx, (ux + dyc).astype(f32), (xl - yn).astype(f32)), ux)
        uz = np.where(toe, np.where(bz, (uz - dyc).astype(f32), (yn - zl).astype(f32)), uz)
        z = np.where(toe, np.where(bz, z, zl), z)
        y = np.where(toe, yn, y)
    out = []
    for cf, base in ((K["inv_r"], y), (K["inv_g"], y), (K["inv_b"], z)):
        w = fma32(ux, cf[1], fma32(uz, cf[2], fma32(base, cf[0], G["c1"])))
        out.append(compress(G, w))
    return np.stack(out, -1)


def build_all():
    E, F, G = build_E(), build_F(), build_G()
    return E, F, G


# ---------------------------------------------------------------------------------------------------------------------
def verify(E, F, G, verbose=True):
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(7)
    # E: float32 values around the kink land on the right side; dense accuracy
    c_lo = f32(KINK_E)
    near = c_lo + np.arange(-64, 65, dtype=np.float32) * f32(2.0 ** -28)
    near = np.unique(np.concatenate([near, np.nextafter(near, f32(1))]))
    idx = e_index(E, near)
    assert np.all((near.astype(f64) > KINK_E) == (idx > E["k"])), "kink cell assignment"
    cs = np.concatenate([rng.random(40000, dtype=np.float32), near, np.linspace(0, 1, 4097, dtype=np.float32), (np.arange(256) / 255.0).astype(f32)])
    assert e_index(E, cs).min() >= 0 and e_index(E, cs).max() < E["n"]
    want = np.array([float(mp.mpf(float(c)) / mp.mpf("12.92") if float(c) <= KINK_E else ((mp.mpf(float(c)) + mp.mpf("0.055")) / mp.mpf("1.055")) ** mp.mpf("2.4")) for c in cs])
    got = expand(E, cs).astype(f64)
    ulp = f32_grid(want)
    err_e = (np.abs(got - want) / ulp)[cs > KINK_E].max()          # in float32 ulps of the result (power branch)
    assert (np.abs(got - want) / ulp).max() <= 1.0 and np.abs(got - want)[cs <= KINK_E].max() < 2e-10
    assert expand(E, f32([0.0]))[0] == 0.0
    # F
    vs = np.concatenate([rng.random(30000), np.exp(rng.uniform(np.log(1e-4), 0.0, 30000)), [0.0, 1.0, 0.0088, 0.0089, 0.5]]).astype(f32)
    vs = vs[np.abs(vs.astype(f64) - KINK_F) > 1e-8]
    us = (vs + F["c0"]).astype(f32)
    vv = us.astype(f64) - f64(F["c0"])
    want = np.array([float(mp.cbrt(mp.mpf(float(v)))) if float(v) > KINK_F else float(mp.mpf("7.787") * mp.mpf(float(v)) + mp.mpf(16) / 116) for v in vv])
    a0, r = lab_f_parts(F, us)
    err_f = np.abs(a0.astype(f64) + r.astype(f64) - want).max()
    err_f1 = np.abs(lab_f_value(F, us).astype(f64) - want).max()
    # G
    ws = np.concatenate([rng.random(30000), np.exp(rng.uniform(np.log(1e-5), 0.0, 30000)), [0.0, 1.0, 0.0031, 0.0032]]).astype(f32)
    ws = ws[np.abs(ws.astype(f64) - KINK_G) > 1e-8]
    wsh = (ws + G["c1"]).astype(f32)
    uu = np.clip(wsh.astype(f64) - f64(G["c1"]), 0.0, None)
    want = np.array([float(mp.mpf("12.92") * mp.mpf(float(u))) if float(u) <= KINK_G else float(mp.mpf("1.055") * mp.mpf(float(u)) ** (1 / mp.mpf("2.4")) - mp.mpf("0.055")) for u in uu])
    want = np.clip(want, 0, 1)
    got = compress(G, wsh).astype(f64)
    err_g = np.abs(got - want).max()
    ends = compress(G, f32([-1.0, 0.0, float(G["c1"]), float(G["whi"]), 2.0]))
    assert ends[0] == 0 and ends[1] == 0 and ends[2] == 0 and ends[3] == 1 and ends[4] == 1
    assert got.min() >= 0 and got.max() <= 1
    if verbose:
        print("table E: S=%.9g k=%d n=%d  max err %.3f float32 ulp of the result" % (E["S"], E["k"], E["n"], err_e))
        print("table F: n=%d c0=%.10g  a0 + r max abs err %.3g, rounded float32 value max abs err %.3g" % (F["n"], F["c0"], err_f, err_f1))
        print("table G: n=%d c1=%.10g  max abs err %.3g (float32 half-ulp at 1 is 3e-8)" % (G["n"], G["c1"], err_g))
    assert err_e < 0.65            # a0 carries no rounding: within 0.15 ulp of correctly rounded (the cells next to the kink), 1 ulp on the linear segment
    assert err_f < 2.5e-9 and err_f1 < 3.3e-8
    assert err_g < 3.6e-8
    return err_e, err_f, err_g


def hexf(x):
    return float(x).hex() + "f"


def render(E, F, G):
    K = consts(F, G)
    L = []
    w = L.append
    w("// ct_lab_tables.h -- GENERATED by tools/gen_lab_tables.py (do not edit; `python tools/gen_lab_tables.py` rewrites it).")
    w("// Look-up tables and constants of the float32 sRGB <-> Lab path (ct_color_lut.h).  Values are hex float literals, exact.")
    w("#pragma once")
    w("#include <stdint.h>")
    w("namespace ct { namespace lut {")
    w("struct alignas(16) Entry { float a0, a1, a2, node; };   // value = a0 + d * (a1 + d * a2), d = x - node")
    w("constexpr float kEScale = %s;          // %.9g: cell boundary %d.46875 sits on the 0.04045 kink" % (hexf(E["S"]), E["S"], E["k"]))
    w("constexpr float kEMagic = %s;          // 1.5 * 2^19 + 0.5: bits(fmaf(c, kEScale, kEMagic)) & kEMask = 16 * cell" % hexf(E_MAGIC))
    w("constexpr uint32_t kEMask = 0x%xu;" % E_MASK)
    w("constexpr int kEEntries = %d;" % E["n"])
    w("constexpr int kFBits = %d;" % F_BITS)
    w("constexpr int kFEntries = %d;" % F["n"])
    w("constexpr float kFShift = %s;          // c0 = %.10g: u = v + c0" % (hexf(F["c0"]), F["c0"]))
    w("constexpr float kFKink = %s;           // u of the 0.008856 kink of f(): a cell boundary" % hexf(F["ukink"]))
    w("constexpr uint32_t kFKinkBits = 0x%08xu;" % int(bits_of(F["ukink"])))
    w("constexpr int kGBits = %d;" % G_BITS)
    w("constexpr int kGEntries = %d;" % G["n"])
    w("constexpr int kGFirst = %d;            // (exponent & 15) * 128 of the first cell" % G["first"])
    w("constexpr float kGShift = %s;          // c1 = %.10g: w = u + c1" % (hexf(G["c1"]), G["c1"]))
    w("constexpr float kGHi = %s;             // 1 + c1: w is clamped to [kGShift, kGHi]" % hexf(G["whi"]))
    for k, name in enumerate("XYZ"):
        o = ROW_ORDER[k]
        w("constexpr float kM%s[3] = {%s, %s, %s};   // row %s / white, in accumulation order: channels %d, %d, %d" % (
            name, hexf(f32(M[k, o[0]])), hexf(f32(M[k, o[1]])), hexf(f32(M[k, o[2]])), name, o[0], o[1], o[2]))
    w("constexpr int kOrd[3][3] = {{%d, %d, %d}, {%d, %d, %d}, {%d, %d, %d}};" % tuple(c for o in ROW_ORDER for c in o))
    w("constexpr float kInvR[3] = {%s, %s, %s};   // lin r = [0] y + [1] (x - y) + [2] (y - z)" % tuple(hexf(v) for v in K["inv_r"]))
    w("constexpr float kInvG[3] = {%s, %s, %s};   // lin g = [0] y + [1] (x - y) + [2] (y - z)" % tuple(hexf(v) for v in K["inv_g"]))
    w("constexpr float kInvB[3] = {%s, %s, %s};   // lin b = [0] z + [1] (x - y) + [2] (y - z)" % tuple(hexf(v) for v in K["inv_b"]))
    w("constexpr float kToeInv = %s;          // 0.2068966" % hexf(K["toe_inv"]))
    w("constexpr float kToeA = %s, kToeB = %s;   // (t - 16/116) / 7.787 = t kToeA + kToeB" % (hexf(K["toe_a"]), hexf(K["toe_b"])))
    w("")
    for name, T in (("kTableE", E), ("kTableF", F), ("kTableG", G)):
        w("__device__ const Entry %s[%d] = {" % (name, T["n"]))
        for e in T["tab"]:
            w("  {%s, %s, %s, %s}," % tuple(hexf(x) for x in e))
        w("};")
    w("}}  // namespace ct::lut")
    return "\n".join(L) + "\n"


def main():
    tabs = build_all()
    verify(*tabs)
    text = render(*tabs)
    if "--check" in sys.argv:
        same = os.path.exists(HEADER) and open(HEADER).read() == text
        print("header up to date" if same else "HEADER DIFFERS from the generator output")
        sys.exit(0 if same else 1)
    open(HEADER, "w").write(text)
    print("wrote %s (%d bytes)" % (HEADER, len(text)))


if __name__ == "__main__":
    main()
