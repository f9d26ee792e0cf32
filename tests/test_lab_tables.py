"""CPU checks of the look-up tables behind the float32 Lab path (csrc/ct_lab_tables.h, csrc/ct_color_lut.h): the committed header
is what tools/gen_lab_tables.py generates, the generator's numpy emulation of the device arithmetic meets its accuracy bounds
against 40-digit arithmetic (mpmath), and the numpy MODEL of the whole float32 pipeline (forward transform, affine map, inverse
transform, rounding for rounding as the kernels do it) stays inside the Lab error budget against the float64 oracle."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mp = pytest.importorskip("mpmath")

from oracle import lab as olab       # noqa: E402


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_lab_tables", os.path.join(ROOT, "tools", "gen_lab_tables.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def tables(gen):
    return gen.build_all()


def test_header_is_generated(gen, tables):
    assert open(gen.HEADER).read() == gen.render(*tables), "run `python tools/gen_lab_tables.py`"


def test_table_accuracy_bounds(gen, tables):
    err_e, err_f, err_g = gen.verify(*tables, verbose=False)
    assert err_e < 0.65         # gamma expansion: float32 ulps of the result (0.5 = correctly rounded)
    assert err_f < 2.5e-9       # f(): grid part + remainder, absolute
    assert err_g < 3.6e-8       # gamma compression, absolute (float32 half-ulp at 1 is 3e-8)


def test_constants_match_the_oracle(gen):
    np.testing.assert_array_equal(gen.XYZ_FROM_RGB, olab.XYZ_FROM_RGB)
    np.testing.assert_array_equal(gen.WHITE, olab.WHITE_D65)
    assert (gen.KINK_E, gen.KINK_F, gen.KINK_FI, gen.KINK_G) == (0.04045, 0.008856, 0.2068966, 0.0031308)


def test_kink_sits_on_a_cell_boundary(gen, tables):
    """no float32 input can land in a table cell that straddles the 0.04045 kink of the sRGB transfer function"""
    E = tables[0]
    c = np.float32(0.04045)
    xs = np.array([np.nextafter(c, np.float32(0)), c, np.nextafter(c, np.float32(1))], np.float32)
    idx = gen.e_index(E, xs)
    assert [int(i) for i in idx] == [E["k"] if float(x) <= 0.04045 else E["k"] + 1 for x in xs]


def test_grid_parts_of_f_are_exact_differences(gen, tables):
    """a0 of table F is a multiple of 2^-24 below 1: the float32 difference of any two is exact"""
    a0 = tables[1]["tab"][:, 0].astype(np.float64)
    assert np.all(a0 * 2 ** 24 == np.rint(a0 * 2 ** 24)) and a0.max() < 1 and a0.min() > 0
    rng = np.random.default_rng(0)
    i, j = rng.integers(0, len(a0), 2000), rng.integers(0, len(a0), 2000)
    d32 = tables[1]["tab"][i, 0] - tables[1]["tab"][j, 0]
    assert np.array_equal(d32.astype(np.float64), a0[i] - a0[j])


def _special(rng, n):
    u = rng.random((n, 3), dtype=np.float32)
    yield "uniform", u, rng.random((n, 3), dtype=np.float32)
    yield "u8", (rng.integers(0, 256, (n, 3)).astype(np.float32) / 255), (rng.integers(0, 256, (n, 3)).astype(np.float32) / 255)
    yield "dark", (u * np.float32(0.12)), (u[::-1] * np.float32(0.2))
    yield "wide", (u * np.float32(0.5) + np.float32(0.25)), u          # scale ~1.9: the forward error doubles


def test_model_of_the_float32_pipeline_vs_oracle(gen, tables):
    """forward (apply grade) + affine + inverse, as the kernels compute them, against methods/linear.py:25-40 in float64.
    Pixels the kernels send through the exact code (within rounding of a kink of Lab's f(), where the reference function jumps by
    1.7e-4 in a*) are excluded exactly as the kernels exclude them."""
    E, F, G = tables
    K = gen.consts(F, G)
    rng = np.random.default_rng(11)
    for name, t, r in _special(rng, 400000):
        lt, lr = olab.rgb2lab(t.astype(np.float64)), olab.rgb2lab(r.astype(np.float64))
        mt, sdt, mr, sdr = lt.mean(0), lt.std(0), lr.mean(0), lr.std(0)
        lab_ref = (lt - mt) * (sdr / sdt) + mr
        rgb_ref = olab.lab2rgb(lab_ref)
        fy, dxy, dyz = gen.forward_apply(E, F, t)
        sy, sxy, syz = gen.forward_stats(E, F, t)
        ry, rxy, ryz = gen.forward_stats(E, F, r)
        st = np.stack([sy, sxy, syz], -1).astype(np.float64)
        sr = np.stack([ry, rxy, ryz], -1).astype(np.float64)
        # the statistics sweep: plain float32 values, unbiased -> the means agree with the oracle's to ~1e-6 Lab
        scale = np.array([116.0, 500.0, 200.0])
        assert np.abs(st.mean(0) * scale - np.array([16.0, 0, 0]) - mt).max() < 3e-6, name
        assert np.abs(st.std(0) * scale - sdt).max() < 3e-6, name
        sc = sr.std(0) / st.std(0)
        off = sr.mean(0) - sc * st.mean(0)
        f32 = np.float32
        gy = gen.fma32(fy, f32(sc[0]), f32(off[0]))
        dx = gen.fma32(dxy, f32(sc[1]), f32(off[1]))
        dz = gen.fma32(dyz, f32(sc[2]), f32(off[2]))
        lab_dev = np.stack([gen.fma32(gy, f32(116), f32(-16)), f32(500) * dx, f32(200) * dz], -1).astype(np.float64)
        out = gen.inverse(G, K, gy, dx, dz)
        # kink bands (ct_color_lut.h: kFBand, kInvBand)
        lin = gen.lin_exact(t.astype(np.float64))
        v = lin @ gen.M.T
        g = np.stack([gy + dx, gy, gy - dz], -1).astype(np.float64)
        keep = (np.abs(v - gen.KINK_F).min(1) > 1.2e-8) & (np.abs(g - gen.KINK_FI).min(1) > 4e-7)
        assert keep.mean() > 0.999
        e_lab = np.abs(lab_dev - lab_ref)[keep].max()
        e_rgb = np.abs(out - rgb_ref)[keep].max()
        e_lab_rgb = np.abs(olab.rgb2lab(out.astype(np.float64)) - olab.rgb2lab(rgb_ref))[keep].max()
        assert out.min() >= 0 and out.max() <= 1
        assert e_lab < 4e-5 and e_lab_rgb < 5e-5 and e_rgb < 6e-6, (name, e_lab, e_lab_rgb, e_rgb)
