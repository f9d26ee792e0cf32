#!/usr/bin/env python3
"""Generates color-transfer_amd/csrc/ct_lab_tables.h: the three look-up tables of the table-driven
sRGB <-> Lab path (csrc/ct_color_lut.h) and the constants that go with them.

The functions tabulated are the scalar pieces of scikit-image 0.18.3's rgb2lab / lab2rgb as the
reference calls them (methods/linear.py:25,26,40):

  A  sRGB gamma expansion   c -> c/12.92 (c <= 0.04045) | ((c+0.055)/1.055)**2.4        c in [0,1]
     uniform grid, index = round(c*S); S is chosen so that a grid boundary sits exactly between the two
     float32 neighbours of 0.04045 (no segment straddles the kink).  Entry = {double a0; float a1, a2}:
     value = a0 + d*(a1 + d*a2), d = c - idx/S.
  B  r = v**(-1/3) at the nodes of a log grid (256 nodes per octave, v in [2^-7, 2)), doubles.
     cbrt(v) = b*g(e) with b = v r^2, e = b r = v r^3 in 1 +- 2^-9 and g(e) = e**(-2/3) as ONE quadratic.
  A32, B32  float32 images of A and of the cube root for the statistics sweep (which only needs unbiased per-pixel values):
     {a0, a1, a2, node} and {c, s1, s2, node}: value = c + d*(s1 + d*s2), d = x - node; B32 has 128 nodes per octave.
  C  sRGB gamma compression u -> 1.055*u**(1/2.4) - 0.055 on a log grid (32 nodes per octave, u in [2^-9, 1]),
     float32 cubics {a0,a1,a2,a3} in d = u - node (the result is rounded to float32 anyway).

Everything is evaluated with mpmath at 40 digits; every table is verified here against mpmath on a dense sample,
emulating the device arithmetic (float32 / float64 roundings) with numpy.  `--check` re-generates and compares with
the committed header (used by tests/test_lab_tables.py).
"""
import os
import struct
import sys
from fractions import Fraction

import mpmath as mp
import numpy as np

mp.mp.dps = 40
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "color-transfer_amd", "csrc", "ct_lab_tables.h")

KINK_A = 0.04045          # skimage rgb2xyz (colorconv.py l.657)
KINK_C = 0.0031308        # skimage xyz2rgb (colorconv.py l.615)
MAGIC = np.float32(12582912.0)   # 1.5 * 2^23: float32 ulp there is 1.0

B_BITS = 8                # table B: mantissa bits per octave (256 nodes)
B_EXP0 = 1016             # biased double exponent of 2^-7; 8 octaves -> v in [2^-7, 2)
B32_BITS = 7              # table B32: 128 nodes per octave (float32 statistics sweep)
C_BITS = 5                # table C: 32 nodes per octave
C_EXP0 = 112              # table C is addressed with (exp & 15): 16 octaves from biased float exponent 112 (2^-15)
C_EXP_FIRST = 118         # first octave that is actually filled: 2^-9 <= 0.0031308


def f32(x):
    return np.float32(x)


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def srgb_expand(c):
    c = mp.mpf(c)
    return c / mp.mpf("12.92") if c <= mp.mpf(KINK_A) else ((c + mp.mpf("0.055")) / mp.mpf("1.055")) ** mp.mpf("2.4")


def srgb_compress_pow(u):
    return mp.mpf("1.055") * mp.mpf(u) ** (1 / mp.mpf("2.4")) - mp.mpf("0.055")


# ---------------------------------------------------------------------------------------------------------------------
# table A
# ---------------------------------------------------------------------------------------------------------------------
def choose_scale():
    """float32 S with  c_lo*S < k+0.5 < c_hi*S  for the float32 neighbours c_lo <= 0.04045 < c_hi (exact rationals):
    fmaf(c, S, MAGIC) then rounds every float32 c <= 0.04045 to index <= k and every c > 0.04045 to index >= k+1."""
    c_lo = np.float32(KINK_A)
    if float(c_lo) > KINK_A:
        c_lo = np.nextafter(c_lo, np.float32(0))
    c_hi = np.nextafter(c_lo, np.float32(1))
    assert float(c_lo) <= KINK_A < float(c_hi)
    for k in (41, 40, 42, 39, 43, 38, 44, 45, 37, 46, 36):
        s0 = np.float32((k + 0.5) / KINK_A)
        for s in (s0, np.nextafter(s0, np.float32(0)), np.nextafter(s0, np.float32(1e9))):
            lo = Fraction(float(c_lo)) * Fraction(float(s))
            hi = Fraction(float(c_hi)) * Fraction(float(s))
            if lo < Fraction(2 * k + 1, 2) < hi:
                return float(s), k
    raise RuntimeError("no float32 scale puts a grid boundary on the sRGB kink")


def cheb_nodes(lo, hi, n):
    return [(lo + hi) / 2 + (hi - lo) / 2 * mp.cos(mp.pi * (2 * j + 1) / (2 * n)) for j in range(n)]


def interp_poly(f, lo, hi, deg):
    """coefficients (low order first) of the degree-`deg` interpolant of f at the Chebyshev nodes of [lo, hi]"""
    xs = cheb_nodes(mp.mpf(lo), mp.mpf(hi), deg + 1)
    A = mp.matrix([[x ** p for p in range(deg + 1)] for x in xs])
    b = mp.matrix([f(x) for x in xs])
    return list(mp.lu_solve(A, b))


def build_table_a():
    S, k = choose_scale()
    inv = float(np.float32(1.0 / S))
    n = int(round(S)) + 2
    a0 = np.zeros(n, np.float64)
    a1 = np.zeros(n, np.float32)
    a2 = np.zeros(n, np.float32)
    h = 0.502 / S
    for i in range(n):
        ci = mp.mpf(i) * mp.mpf(inv)                 # the node the device arithmetic implies: d = fmaf(idx, -inv, c)
        if i <= k:
            a0[i] = float(ci / mp.mpf("12.92"))
            a1[i] = np.float32(1.0 / 12.92)
            a2[i] = 0.0
            continue
        f = lambda d: ((ci + d + mp.mpf("0.055")) / mp.mpf("1.055")) ** mp.mpf("2.4")   # pow branch, also left of the kink
        c = interp_poly(f, -h, h, 2)
        a1[i] = np.float32(float(c[1]))
        a2[i] = np.float32(float(c[2]))
        # re-centre a0 for the rounded a1, a2
        errs = [f(d) - (mp.mpf(float(a1[i])) * d + mp.mpf(float(a2[i])) * d * d) for d in mp.linspace(-h, h, 17)]
        a0[i] = float((max(errs) + min(errs)) / 2)
    return dict(S=S, inv=inv, k=k, n=n, a0=a0, a1=a1, a2=a2)


def emulate_a(tab, c32):
    """numpy emulation of the device arithmetic of table A for float32 inputs -> float64 linear values"""
    c32 = np.asarray(c32, np.float32)
    y = (c32.astype(np.float64) * np.float64(np.float32(tab["S"])) + np.float64(MAGIC)).astype(np.float32)   # fmaf
    idx = (y.view(np.uint32).astype(np.int64) - f32_bits(MAGIC))
    yb = (y - MAGIC).astype(np.float32)
    d = (c32.astype(np.float64) - yb.astype(np.float64) * np.float64(np.float32(tab["inv"]))).astype(np.float32)  # fmaf
    t = (d.astype(np.float64) * tab["a2"][idx].astype(np.float64) + tab["a1"][idx].astype(np.float64)).astype(np.float32)
    corr = (d * t).astype(np.float32)
    return tab["a0"][idx] + corr.astype(np.float64), idx


# ---------------------------------------------------------------------------------------------------------------------
# table B
# ---------------------------------------------------------------------------------------------------------------------
def build_table_b():
    n = 8 << B_BITS
    r = np.zeros(n, np.float64)
    for i in range(n):
        e = B_EXP0 + (i >> B_BITS) - 1023
        v = mp.mpf(2) ** e * (1 + mp.mpf(i & ((1 << B_BITS) - 1)) / (1 << B_BITS))
        r[i] = float(v ** (-mp.mpf(1) / 3))
    w = mp.mpf(2) ** -(B_BITS + 1) * mp.mpf("1.02")
    q = interp_poly(lambda e: mp.mpf(e) ** (-mp.mpf(2) / 3), 1 - w, 1 + w, 2)
    return dict(n=n, r=r, q=[float(x) for x in q])


def emulate_b(tab, v):
    v = np.asarray(v, np.float64)
    hi = (v.view(np.uint64) >> np.uint64(32)).astype(np.int64)
    idx = ((hi + (1 << (19 - B_BITS))) >> (20 - B_BITS)) & ((8 << B_BITS) - 1)
    r = tab["r"][idx]
    t = v * r
    b = t * r
    e = b * r
    q0, q1, q2 = tab["q"]
    g = (q2 * e + q1) * e + q0          # two fma; the extra rounding of this emulation is below 1e-16
    return b * g


def build_table_a32(ta):
    """float32 image of table A: the node index / S is rounded to float32 and a0 moves with it (a0 + a1 * shift)"""
    n = ta["n"]
    out = np.zeros((n, 4), np.float32)
    for i in range(n):
        node = mp.mpf(i) * mp.mpf(ta["inv"])
        nodef = np.float32(float(node))
        shift = mp.mpf(float(nodef)) - node
        a0 = mp.mpf(float(ta["a0"][i])) + mp.mpf(float(ta["a1"][i])) * shift + mp.mpf(float(ta["a2"][i])) * shift * shift
        a1 = mp.mpf(float(ta["a1"][i])) + 2 * mp.mpf(float(ta["a2"][i])) * shift
        out[i] = [np.float32(float(a0)), np.float32(float(a1)), ta["a2"][i], nodef]
    return out


def build_table_b32():
    n = 8 << B32_BITS
    out = np.zeros((n, 4), np.float32)
    for i in range(n):
        e = 120 + (i >> B32_BITS) - 127
        m = i & ((1 << B32_BITS) - 1)
        v = mp.mpf(2) ** e * (1 + mp.mpf(m) / (1 << B32_BITS))
        step = mp.mpf(2) ** e / (1 << B32_BITS)
        lo = -step / 2 if m else -step / 4
        q = interp_poly(lambda d: mp.cbrt(v + d), lo * mp.mpf("1.01"), step / 2 * mp.mpf("1.01"), 2)
        out[i] = [np.float32(float(q[0])), np.float32(float(q[1])), np.float32(float(q[2])), np.float32(float(v))]
    return out


def fma32(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def emulate_a32(ta, a32, c32):
    c32 = np.asarray(c32, np.float32)
    y = fma32(c32, np.float32(ta["S"]), MAGIC)
    idx = y.view(np.uint32).astype(np.int64) - f32_bits(MAGIC)
    e = a32[idx]
    d = (c32 - e[:, 3]).astype(np.float32)
    return fma32(d, fma32(d, e[:, 2], e[:, 1]), e[:, 0])


def emulate_b32(b32, v32):
    v32 = np.asarray(v32, np.float32)
    bits = v32.view(np.uint32).astype(np.int64) + (1 << (22 - B32_BITS))
    idx = (bits >> (23 - B32_BITS)) & ((8 << B32_BITS) - 1)
    e = b32[idx]
    d = (v32 - e[:, 3]).astype(np.float32)
    return fma32(d, fma32(d, e[:, 2], e[:, 1]), e[:, 0])


# ---------------------------------------------------------------------------------------------------------------------
# table C
# ---------------------------------------------------------------------------------------------------------------------
def build_table_c():
    n = 16 << C_BITS
    a = np.zeros((n, 4), np.float32)
    first = (C_EXP_FIRST - C_EXP0) << C_BITS
    last = (127 - C_EXP0) << C_BITS              # node u = 1.0
    for i in range(first, last + 1):
        e = C_EXP0 + (i >> C_BITS) - 127
        m = i & ((1 << C_BITS) - 1)
        u = mp.mpf(2) ** e * (1 + mp.mpf(m) / (1 << C_BITS))
        step = mp.mpf(2) ** e / (1 << C_BITS)
        lo = -step / 2 if m else -step / 4          # below a power of two the grid is twice as fine
        hi = step / 2
        if i == last:
            hi = step / 64                          # inputs are clamped to <= 1
        f = lambda d: srgb_compress_pow(u + d)
        c = interp_poly(f, lo * mp.mpf("1.01"), hi * mp.mpf("1.01"), 3)
        a[i, 1:] = [np.float32(float(x)) for x in c[1:]]
        if i == last:
            a[i, 0] = 1.0                            # g(1) = 1 exactly: the clamp at 1 needs no second clip
        else:
            a[i, 0] = np.float32(float(c[0]))
    return dict(n=n, a=a, first=first, last=last)


def emulate_c(tab, u32):
    """float32 emulation of the device evaluation: clamp to [0,1], toe select, cubic in d = u - node"""
    u = np.clip(np.asarray(u32, np.float32), np.float32(0), np.float32(1))
    bits = u.view(np.uint32).astype(np.int64) + (1 << (22 - C_BITS))
    idx = (bits >> (23 - C_BITS)) & ((16 << C_BITS) - 1)
    node = (bits & ~((1 << (23 - C_BITS)) - 1)).astype(np.uint32).view(np.float32)
    d = (u - node).astype(np.float32)
    a = tab["a"][idx]
    p = (d.astype(np.float64) * a[:, 3] + a[:, 2]).astype(np.float32)
    p = (d.astype(np.float64) * p + a[:, 1]).astype(np.float32)
    p = (d.astype(np.float64) * p + a[:, 0]).astype(np.float32)
    lin = (u * np.float32(12.92)).astype(np.float32)
    return np.where(u <= np.float32(KINK_C), lin, p)


# ---------------------------------------------------------------------------------------------------------------------
def verify(ta, tb, tc, a32=None, b32=None, verbose=True):
    rng = np.random.default_rng(7)
    # A: all float32 values around the kink must land on the right side; dense accuracy check
    c_lo = np.float32(KINK_A)
    near = c_lo + np.arange(-64, 65, dtype=np.float32) * np.float32(2.0 ** -28)
    near = np.unique(np.concatenate([near, np.nextafter(near, np.float32(1))]))
    _, idx = emulate_a(ta, near)
    assert np.all((near.astype(np.float64) > KINK_A) == (idx > ta["k"])), "kink segment assignment"
    cs = np.concatenate([rng.random(40000, dtype=np.float32), near, np.linspace(0, 1, 4097, dtype=np.float32),
                         (np.arange(256) / 255.0).astype(np.float32)])
    got, idx = emulate_a(ta, cs)
    assert idx.min() >= 0 and idx.max() < ta["n"]
    want = np.array([float(srgb_expand(float(c))) for c in cs])
    err_a = np.abs(got - want)
    rel_a = (err_a / np.maximum(want, 1e-300))[cs > 0.05].max()
    # B
    vs = np.concatenate([np.exp(rng.uniform(np.log(0.008856), np.log(1.0), 40000)), [1.0, 0.008856, 0.0088561, 0.5, 0.25]])
    got = emulate_b(tb, vs)
    want = np.array([float(mp.cbrt(mp.mpf(float(v)))) for v in vs])
    rel_b = np.abs(got / want - 1).max()
    # C
    us = np.concatenate([np.exp(rng.uniform(np.log(0.002), 0.0, 40000)), [1.0, 0.0031308, 0.0031309, 2.0 ** -9, 0.0, 1e-5, 0.5]]).astype(np.float32)
    got = emulate_c(tc, us)
    want = np.array([float(mp.mpf("12.92") * mp.mpf(float(u)) if float(u) <= KINK_C else srgb_compress_pow(float(u))) for u in us])
    err_c = np.abs(got.astype(np.float64) - want).max()
    assert emulate_c(tc, np.float32([1.0, 2.0]))[0] == 1.0
    if a32 is not None:
        c = np.concatenate([rng.random(40000, dtype=np.float32), (np.arange(256) / 255.0).astype(np.float32)])
        want = np.array([float(srgb_expand(float(x))) for x in c])
        rel_a32 = np.abs(emulate_a32(ta, a32, c).astype(np.float64) - want) / np.maximum(want, 1e-30)
        vv = np.exp(rng.uniform(np.log(0.008856), 0.0, 40000)).astype(np.float32)
        want = np.array([float(mp.cbrt(mp.mpf(float(x)))) for x in vv])
        rel_b32 = np.abs(emulate_b32(b32, vv).astype(np.float64) / want - 1)
        if verbose:
            print("tables A32 / B32: max rel err %.3g / %.3g, mean signed rel err %.2g / %.2g (float32 ulp 6e-8 .. 1.2e-7)" % (
                rel_a32.max(), rel_b32.max(), ((emulate_a32(ta, a32, c).astype(np.float64) - np.array([float(srgb_expand(float(x))) for x in c])) / np.maximum(np.array([float(srgb_expand(float(x))) for x in c]), 1e-30)).mean(),
                (emulate_b32(b32, vv).astype(np.float64) / want - 1).mean()))
        assert rel_a32[c > 1e-3].max() < 2.5e-7 and rel_b32.max() < 1.5e-7
    if verbose:
        print("table A: S=%.9g k=%d n=%d   max abs err %.3g, max rel err (c>0.05) %.3g" % (ta["S"], ta["k"], ta["n"], err_a.max(), rel_a))
        print("table B: n=%d  cbrt max rel err %.3g   q=%s" % (tb["n"], rel_b, tb["q"]))
        print("table C: n=%d (filled %d..%d)  max abs err %.3g (float32 half-ulp at 1 is 3e-8)" % (tc["n"], tc["first"], tc["last"], err_c))
    assert err_a.max() < 3e-10 and rel_a < 1e-8      # quadratic interpolation (dark end) and the float32 d*(a1 + d*a2) term (bright end)
    assert rel_b < 2e-9
    assert err_c < 1.3e-7
    return err_a.max(), rel_b, err_c


def hexd(x):
    return float(x).hex()


def render(ta, tb, tc, a32, b32):
    L = []
    w = L.append
    w("// ct_lab_tables.h -- GENERATED by tools/gen_lab_tables.py (do not edit; `python tools/gen_lab_tables.py` rewrites it).")
    w("// Look-up tables of the table-driven sRGB <-> Lab path (ct_color_lut.h).  Values are hex float literals, exact.")
    w("#pragma once")
    w("#include <stdint.h>")
    w("namespace ct { namespace lut {")
    w("constexpr float kMagic = %sf;            // 1.5 * 2^23" % hexd(MAGIC))
    w("constexpr uint32_t kMagicBits = 0x%08xu;" % f32_bits(MAGIC))
    w("constexpr float kAScale = %sf;          // %.9g: index = round(c * kAScale); boundary %d.5 sits on the 0.04045 kink" % (hexd(np.float32(ta["S"])), ta["S"], ta["k"]))
    w("constexpr float kANegInv = %sf;         // -float(1/kAScale): d = fmaf(index, kANegInv, c)" % hexd(-np.float32(ta["inv"])))
    w("constexpr int kAEntries = %d;" % ta["n"])
    w("constexpr int kBBits = %d;" % B_BITS)
    w("constexpr int kBEntries = %d;" % tb["n"])
    w("constexpr double kBQ0 = %s, kBQ1 = %s, kBQ2 = %s;   // g(e) = e^(-2/3) on 1 +- 2^-%d" % (hexd(tb["q"][0]), hexd(tb["q"][1]), hexd(tb["q"][2]), B_BITS + 1))
    w("constexpr int kCBits = %d;" % C_BITS)
    w("constexpr int kCEntries = %d;          // addressed with (exponent & 15): entries below %d are never read" % (tc["last"] + 1, tc["first"]))
    w("struct alignas(16) EntryA { double a0; float a1, a2; };")
    w("struct alignas(16) EntryC { float a0, a1, a2, a3; };")
    w("struct alignas(16) Entry32 { float c, s1, s2, node; };   // value = c + d * (s1 + d * s2), d = x - node")
    w("constexpr int kB32Bits = %d;" % B32_BITS)
    w("constexpr int kB32Entries = %d;" % len(b32))
    w("")
    w("__device__ const EntryA kTableA[kAEntries] = {")
    for i in range(ta["n"]):
        w("  {%s, %sf, %sf}," % (hexd(ta["a0"][i]), hexd(ta["a1"][i]), hexd(ta["a2"][i])))
    w("};")
    w("__device__ const double kTableB[kBEntries] = {")
    for i in range(0, tb["n"], 4):
        w("  " + " ".join("%s," % hexd(x) for x in tb["r"][i:i + 4]))
    w("};")
    w("__device__ const EntryC kTableC[kCEntries] = {")
    for i in range(tc["last"] + 1):
        w("  {%sf, %sf, %sf, %sf}," % tuple(hexd(x) for x in tc["a"][i]))
    w("};")
    w("__device__ const Entry32 kTableA32[kAEntries] = {")
    for i in range(len(a32)):
        w("  {%sf, %sf, %sf, %sf}," % tuple(hexd(x) for x in a32[i]))
    w("};")
    w("__device__ const Entry32 kTableB32[kB32Entries] = {")
    for i in range(len(b32)):
        w("  {%sf, %sf, %sf, %sf}," % tuple(hexd(x) for x in b32[i]))
    w("};")
    w("}}  // namespace ct::lut")
    return "\n".join(L) + "\n"


def build_all():
    ta = build_table_a()
    return ta, build_table_b(), build_table_c(), build_table_a32(ta), build_table_b32()


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
