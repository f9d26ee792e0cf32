"""Pin the CPU oracle (oracle/) against golden vectors produced by the real reference
(tests/golden/make_golden_linear.py, reference called with float64 inputs)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import lab as olab
from oracle import linear as olin


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def synth_pair(seed, h, w):
    rng = np.random.default_rng(seed)
    return rng.random((h, w, 3), dtype=np.float32), rng.random((h, w, 3), dtype=np.float32)


@pytest.mark.parametrize("case", ["uniform", "graded"])
def test_linear_small_full(golden_dir, case):
    g = _load(golden_dir, "linear_small.npz")
    t, r = g[case + "/target"], g[case + "/reference"]
    mt, st = olin.lab_stats(t)
    mr, sr = olin.lab_stats(r)
    np.testing.assert_allclose(mt, g[case + "/lab_mean_t"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(st, g[case + "/lab_std_t"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(mr, g[case + "/lab_mean_r"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(sr, g[case + "/lab_std_r"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(olin.reinhard_lab(t, r), g[case + "/reinhard_lab"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(olin.color_transfer_between_images(t, r), g[case + "/reinhard"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(olin.color_transfer_in_correlated_color_space(t, r), g[case + "/xiao"],
                               rtol=0, atol=1e-9)
    for d in ("MK", "sqrt", "cholesky"):
        np.testing.assert_allclose(olin.monge_kantorovitch_color_transfer(t, r, decomposition=d),
                                   g[case + "/mk_" + d], rtol=0, atol=1e-10)
    m, c = olin.rgb_mean_cov(t)
    np.testing.assert_allclose(m, g[case + "/rgb_mean_t"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(c, g[case + "/rgb_cov_t"], rtol=0, atol=1e-14)


def test_linear_u8_256(golden_dir):
    g = _load(golden_dir, "linear_u8_256.npz")
    t = g["target_u8"].astype(np.float32) / 255
    r = g["reference_u8"].astype(np.float32) / 255
    sl = (slice(None, None, 3), slice(None, None, 3))
    np.testing.assert_allclose(olin.lab_stats(t)[0], g["lab_mean_t"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(olin.lab_stats(r)[1], g["lab_std_r"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(olin.reinhard_lab(t, r)[sl], g["reinhard_lab_s3"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(olin.color_transfer_between_images(t, r)[sl], g["reinhard_s3"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(olin.color_transfer_in_correlated_color_space(t, r)[sl], g["xiao_s3"], rtol=0, atol=1e-9)
    for d in ("MK", "sqrt", "cholesky"):
        np.testing.assert_allclose(olin.monge_kantorovitch_color_transfer(t, r, decomposition=d)[sl],
                                   g["mk_%s_s3" % d], rtol=0, atol=1e-10)


def test_linear_1080p_samples(golden_dir):
    g = _load(golden_dir, "linear_1080p.npz")
    t, r = synth_pair(int(g["seed"]), 1080, 1920)
    assert hashlib.sha256(t.tobytes()).hexdigest() == str(g["target_sha256"]), "numpy Generator stream drifted"
    assert hashlib.sha256(r.tobytes()).hexdigest() == str(g["reference_sha256"])
    idx = g["idx"]
    mt, st = olin.lab_stats(t)
    np.testing.assert_allclose(mt, g["lab_mean_t"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(st, g["lab_std_t"], rtol=0, atol=1e-10)
    out = olin.color_transfer_between_images(t, r).reshape(-1, 3)[idx]
    np.testing.assert_allclose(out, g["reinhard_samples"], rtol=0, atol=1e-12)
    out = olin.monge_kantorovitch_color_transfer(t, r).reshape(-1, 3)[idx]
    np.testing.assert_allclose(out, g["mk_MK_samples"], rtol=0, atol=1e-10)
    out = olin.color_transfer_in_correlated_color_space(t, r).reshape(-1, 3)[idx]
    np.testing.assert_allclose(out, g["xiao_samples"], rtol=0, atol=1e-9)
    # informational: the reference's own float32-input self-noise in Lab is far above 1e-4
    noise = np.abs(olab.rgb2lab(g["reinhard_f32in_samples"].astype(np.float64)) - olab.rgb2lab(g["reinhard_samples"])).max()
    assert noise > 1e-4


def test_lab_roundtrip_property():
    rng = np.random.default_rng(0)
    x = rng.random((10, 13, 3))
    np.testing.assert_allclose(olab.lab2rgb(olab.rgb2lab(x)), x, rtol=0, atol=1e-6)


def test_mk_unknown_decomposition():
    x = np.random.default_rng(0).random((4, 4, 3))
    with pytest.raises(ValueError):
        olin.monge_kantorovitch_color_transfer(x, x, decomposition="nope")
