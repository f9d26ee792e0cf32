"""CPU checks of the look-up tables behind the table-driven Lab path (csrc/ct_lab_tables.h, csrc/ct_color_lut.h):
the committed header is what tools/gen_lab_tables.py generates, and the generator's numpy emulation of the device
arithmetic meets its accuracy bounds against 40-digit arithmetic (mpmath)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mp = pytest.importorskip("mpmath")


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
    err_a, rel_b, err_c = gen.verify(*tables, verbose=False)    # also asserts the float32 tables A32 / B32 (<= 2 ulp)
    assert err_a < 3e-10        # linear values, absolute
    assert rel_b < 2e-9         # cube roots, relative
    assert err_c < 1.3e-7       # gamma compression, absolute (float32 result)


def test_kink_sits_on_a_grid_boundary(gen, tables):
    """no float32 input can land in a table segment that straddles the 0.04045 kink of the sRGB transfer function"""
    ta = tables[0]
    c = np.float32(0.04045)
    xs = np.array([np.nextafter(c, np.float32(0)), c, np.nextafter(c, np.float32(1))], np.float32)
    _, idx = gen.emulate_a(ta, xs)
    assert [int(i) for i in idx] == [ta["k"] if float(x) <= 0.04045 else ta["k"] + 1 for x in xs]


def test_lab_from_tables_vs_oracle(gen, tables):
    """rgb -> Lab through the emulated tables against the float64 oracle on random, 8-bit and dark pixels"""
    from oracle import lab as olab
    ta, tb = tables[0], tables[1]
    rng = np.random.default_rng(3)
    rgb = np.concatenate([rng.random((20000, 3), dtype=np.float32), rng.integers(0, 256, (5000, 3)).astype(np.float32) / 255,
                          rng.random((5000, 3), dtype=np.float32) * np.float32(0.1)])
    lin = np.stack([gen.emulate_a(ta, rgb[:, k])[0] for k in range(3)], 1)
    xyz = lin @ (olab.XYZ_FROM_RGB / olab.WHITE_D65[:, None]).T
    f = np.where(xyz > 0.008856, gen.emulate_b(tb, np.maximum(xyz, 2.0 ** -7)), 7.787 * xyz + 16.0 / 116.0)
    lab = np.stack([116 * f[:, 1] - 16, 500 * (f[:, 0] - f[:, 1]), 200 * (f[:, 1] - f[:, 2])], 1)
    assert np.abs(lab - olab.rgb2lab(rgb.astype(np.float64))).max() < 1e-6
