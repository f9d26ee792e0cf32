"""Regrain / automated_color_grading (reference methods/iterative.py:62-138).

not-gpu: the oracle (oracle/regrain.py, incl. its restatement of skimage.transform.resize) against fixtures produced by the
real reference + real scikit-image 0.18.3 (tests/golden/make_golden_regrain.py);  gpu: ct_regrain_f64 and
methods.iterative.automated_color_grading against those fixtures and the oracle."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import regrain as org        # noqa: E402


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "regrain.npz"), allow_pickle=False)


def test_oracle_vs_reference_run(g):
    for tag in "abc":
        i, c = g[tag + "/in"], g[tag + "/col"]
        h, w = i.shape[:2]
        down = org.resize(i, ((h + 1) // 2, (w + 1) // 2))
        assert np.abs(down - g[tag + "/resize_down"]).max() < 1e-13                  # real skimage.transform.resize, shrinking
        assert np.abs(org.resize(down, (h, w)) - g[tag + "/resize_up"]).max() < 1e-13    # ... and growing
        assert np.abs(org.solve(i, i, c, 4, 0) - g[tag + "/solve_l0_n4"]).max() < 1e-13
        assert np.abs(org.solve(c * 0.5 + i * 0.5, i, c, 7, 2) - g[tag + "/solve_l2_n7"]).max() < 1e-13
        assert np.abs(org.regrain(i, c) - g[tag + "/regrain"]).max() < 1e-13
        # a float32 target: the reference resizes it in float32, the oracle (and the device) in float64
        assert np.abs(org.regrain(i, c) - g[tag + "/regrain_f32in"]).max() < 2e-6


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_regrain_vs_reference_run(g, hip):
    for tag in "abc":
        i, c = g[tag + "/in"], g[tag + "/col"]
        out = hip.regrain(dev(i), dev(c)).cpu().numpy()
        assert out.dtype == np.float64 and np.abs(out - g[tag + "/regrain"]).max() < 1e-12, tag
        out32 = hip.regrain(dev(i.astype(np.float32)), dev(c)).cpu().numpy()
        assert np.abs(out32 - g[tag + "/regrain_f32in"]).max() < 2e-6
        # a single level / a single sweep count exercise _solve alone
        one = hip.regrain(dev(i), dev(c), nbits=(4,)).cpu().numpy()
        assert np.abs(one - g[tag + "/solve_l0_n4"]).max() < 1e-12


@pytest.mark.gpu
def test_automated_color_grading_vs_reference_run(g):
    import methods.iterative as it
    t, r = g["acg/target"], g["acg/reference"]
    out = it.automated_color_grading(t, r, rotations=g["acg/rotations"])
    assert out.dtype == np.float64 and out.shape == t.shape
    assert np.abs(out - g["acg/out"]).max() < 5e-6            # float32 target: the reference's pyramid of it is float32
    np.random.seed(5)                                         # default path: rotations from numpy's global RNG like the reference
    out2 = it.automated_color_grading(t, r)
    assert out2.shape == t.shape and np.isfinite(out2).all()
    from methods import Runner
    batch = {"target": torch.from_numpy(t).permute(2, 0, 1)[None].cuda(), "reference": torch.from_numpy(t).permute(2, 0, 1)[None].cuda()}
    assert Runner("methods.iterative.automated_color_grading")(batch).shape == (1, 3) + t.shape[:2]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1080, 1920), (271, 483), (40, 45), (21, 300)])
def test_hip_regrain_vs_oracle_sizes(hip, shape):
    """full size (six pyramid levels), odd sizes, and sizes where the recursion stops at once"""
    h, w = shape
    rng = np.random.default_rng(h)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    i = np.clip(np.stack([0.5 + 0.4 * np.sin(xx / 15.0), 0.2 + 0.6 * yy / h, 0.5 + 0.4 * np.cos((xx + yy) / 23.0)], -1) + 0.03 * rng.standard_normal((h, w, 3)), 0, 1)
    c = np.clip(i ** 0.9 * 0.95 + 0.03 * rng.standard_normal((h, w, 3)), 0, 1)
    out = hip.regrain(dev(i), dev(c)).cpu().numpy()
    want = org.regrain(i, c)
    assert np.abs(out - want).max() < 1e-11
    assert np.array_equal(out, hip.regrain(dev(i), dev(c)).cpu().numpy())      # deterministic
