"""Pin oracle/idt_oracle.c against goldens captured from the real reference
(tests/golden/make_golden_idt.py): rotations are data, everything else must match --
bin indices / counts exactly on iteration 0, outputs to 1e-9."""
import os

import numpy as np
import pytest

from oracle import iterative as oit


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _moved(h_a, h_b):
    """number of pixels sitting in a different bin = half the L1 distance of the histograms"""
    return int(np.abs(h_a.astype(np.int64) - h_b.astype(np.int64)).sum() // 2)


@pytest.mark.parametrize("case,bins,n_iter", [("f64", 255, 4), ("f32", 255, 4), ("odd", 64, 2)])
def test_idt_small_vs_reference(golden_dir, case, bins, n_iter):
    g = _g(golden_dir, "idt_small.npz")
    t, r = g[case + "/target"], g[case + "/reference"]
    out, dbg = oit.iterative_distribution_transfer(t, r, bins=bins, n_iter=n_iter, rotations=g[case + "/rot"],
                                                   debug=True)
    # iteration 0: everything that is integer is exact
    assert np.array_equal(dbg["lohi"][0], g[case + "/lohi"][0])
    assert np.array_equal(dbg["binidx"][0], g[case + "/binidx"][0])
    assert np.array_equal(dbg["hist0"][0], g[case + "/hist0"][0])
    assert np.array_equal(dbg["hist1"][0], g[case + "/hist1"][0])
    np.testing.assert_allclose(dbg["lut"][0], g[case + "/lut"][0], rtol=0, atol=1e-13)
    # reference histograms never change (same image, same rotation): exact at every iteration
    # up to lo/hi, which depend on the working image
    for it in range(1, n_iter):
        np.testing.assert_allclose(dbg["lohi"][it], g[case + "/lohi"][it], rtol=0, atol=1e-12)
        # later iterations: the working image differs from the reference's by ~1e-16 (LU solve vs
        # inverse), so a handful of pixels sitting on a bin edge may move
        for j in range(3):
            assert _moved(dbg["hist0"][it, j], g[case + "/hist0"][it, j]) <= 2
            assert _moved(dbg["hist1"][it, j], g[case + "/hist1"][it, j]) <= 2
        np.testing.assert_allclose(dbg["lut"][it], g[case + "/lut"][it], rtol=0, atol=1e-9)
    assert out.dtype == np.float64 and out.shape == t.shape
    np.testing.assert_allclose(out, g[case + "/out"], rtol=0, atol=1e-9)


def test_projection_fma_order_matches_numpy_matmul(golden_dir):
    """numpy 2.2.6's `r @ x.T` is the FMA chain the oracle (and the HIP kernel) pins."""
    g = _g(golden_dir, "idt_small.npz")
    t = g["f64/target"].reshape(-1, 3)
    r = g["f64/rot"][0]
    assert np.array_equal(oit.project(t, r), r @ t.T) or np.abs(oit.project(t, r) - r @ t.T).max() < 1e-15


def test_idt_u8_256_vs_reference(golden_dir):
    g = _g(golden_dir, "idt_u8_256.npz")
    t = g["target_u8"].astype(np.float32) / 255
    r = g["reference_u8"].astype(np.float32) / 255
    out, dbg = oit.iterative_distribution_transfer(t, r, rotations=g["rot"], debug=True)
    assert np.array_equal(dbg["lohi"][0], g["lohi"][0])
    assert np.array_equal(dbg["binidx"][0].astype(np.uint8), g["binidx_it0"])
    assert np.array_equal(dbg["hist0"][0], g["hist0"][0])
    assert np.array_equal(dbg["hist1"][0], g["hist1"][0])
    for it in range(1, 4):
        for j in range(3):
            assert _moved(dbg["hist0"][it, j], g["hist0"][it, j]) <= 4
    np.testing.assert_allclose(dbg["lut"], g["lut"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out[::3, ::3], g["out_s3"], rtol=0, atol=1e-9)


def test_idt_1080p_vs_reference(golden_dir):
    g = _g(golden_dir, "idt_1080p.npz")
    rng = np.random.default_rng(int(g["seed"]))
    t = rng.random((1080, 1920, 3), dtype=np.float32)
    r = rng.random((1080, 1920, 3), dtype=np.float32)
    out, dbg = oit.iterative_distribution_transfer(t, r, rotations=g["rot"], debug=True)
    idx = g["idx"]
    assert np.array_equal(dbg["lohi"][0], g["lohi"][0])
    assert np.array_equal(dbg["hist0"][0], g["hist0"][0])
    assert np.array_equal(dbg["hist1"][0], g["hist1"][0])
    assert np.array_equal(dbg["binidx"][0][:, idx].astype(np.uint8), g["binidx_it0_samples"])
    for it in range(1, 4):
        for j in range(3):
            assert _moved(dbg["hist0"][it, j], g["hist0"][it, j]) <= 8
    np.testing.assert_allclose(out.reshape(-1, 3)[idx], g["out_samples"], rtol=0, atol=1e-9)


def test_first_bin_maps_to_zero_quirk():
    """np.interp(..., left=0): every pixel of the first histogram bin gets d_r = 0.0."""
    rng = np.random.default_rng(0)
    t = rng.random((16, 16, 3))
    r = rng.random((16, 16, 3)) * 0.9 + 0.05
    t[0, 0] = 0.0                            # the global minimum of every axis is a target pixel
    rot = np.eye(3)[None]
    out, dbg = oit.iterative_distribution_transfer(t, r, n_iter=1, rotations=rot, debug=True)
    first = dbg["binidx"][0] == 0           # [3, n]
    for j in range(3):
        sel = first[j]
        assert sel.any()
        # identity rotation: new value = old + (d_r - old) = d_r = 0.0 exactly
        assert np.all(out.reshape(-1, 3)[sel, j] == 0.0)
