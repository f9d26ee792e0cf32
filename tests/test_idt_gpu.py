"""GPU parity for iterative_distribution_transfer: HIP kernels vs the C oracle (bitwise: bin
indices, histogram counts, LUTs, output) and vs the reference goldens."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import iterative as oit   # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


@pytest.fixture(scope="module")
def it():
    import methods.iterative as m
    return m


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_hip(hip, t, r, rot, bins=255):
    out, dbg = hip.idt(dev(t), dev(r), rot, bins=bins, debug=True)
    torch.cuda.synchronize()
    d = {k: v.cpu().numpy() for k, v in dbg.items()}
    return out.cpu().numpy(), d


def assert_bitwise_vs_oracle(out, d, t, r, rot, bins):
    o_out, o = oit.iterative_distribution_transfer(t, r, bins=bins, n_iter=rot.shape[0], rotations=rot, debug=True)
    assert np.array_equal(d["par"][0, :, :, 0:2], o["lohi"]), "lo/hi differ"
    assert np.array_equal(d["binidx"][0].astype(np.uint16), o["binidx"]), "bin indices differ"
    assert np.array_equal(d["hist"][0, :, 0].astype(np.int64), o["hist0"]), "target histogram counts differ"
    assert np.array_equal(d["hist"][0, :, 1].astype(np.int64), o["hist1"]), "reference histogram counts differ"
    assert np.array_equal(d["lut"][0, :, :, :, 0], o["lut"]), "LUT f differs"
    assert np.array_equal(out, o_out), "float64 output not bitwise equal to the oracle"
    return o_out


def assert_later_iterations_vs_reference(d, g, case, t, r, rot, bins):
    """Iterations >= 1 against the REAL reference's integers (numpy's own per-pixel bin indices, counts, lo/hi): the working
    image of those iterations differs from the reference's by its rounding (~1e-16 relative: numpy.linalg.solve against our
    inverse x fma chain), so a bin index may differ only where the projected value sits on a bin edge to within 1e-12 of
    the axis range, and then by exactly one bin; lo/hi agree to 1e-12 of the range; a histogram count moves by at most the
    number of such pixels."""
    _, o = oit.iterative_distribution_transfer(t, r, bins=bins, n_iter=rot.shape[0], rotations=rot, debug=True)
    n_edge = 0
    for it in range(1, rot.shape[0]):
        want_idx, got_idx = g[case + "/binidx"][it].astype(np.int64), d["binidx"][0, it].astype(np.int64)
        lohi_ref = g[case + "/lohi"][it]
        lo, hi = d["par"][0, it, :, 0], d["par"][0, it, :, 1]
        span = hi - lo
        np.testing.assert_allclose(d["par"][0, it, :, 0:2], lohi_ref, rtol=0, atol=1e-12 * float(span.max()))
        # the oracle's working image at the start of iteration `it` (bitwise the device's) and its projection
        proj = oit.project(o["state"][it - 1], rot[it])                      # [3, n]
        for ax in range(3):
            bad = np.nonzero(want_idx[ax] != got_idx[ax])[0]
            n_edge += bad.size
            if bad.size == 0:
                continue
            assert np.abs(want_idx[ax][bad] - got_idx[ax][bad]).max() == 1, (it, ax)
            pos = (proj[ax][bad] - lo[ax]) / span[ax] * bins                  # in units of bins
            assert np.abs(pos - np.rint(pos)).max() <= 1e-12 * bins * 4, (it, ax, float(np.abs(pos - np.rint(pos)).max()))
            assert np.abs(d["hist"][0, it, 0, ax].astype(np.int64) - g[case + "/hist0"][it][ax]).sum() <= 2 * bad.size
        if not any((want_idx[ax] != got_idx[ax]).any() for ax in range(3)):
            assert np.array_equal(d["hist"][0, it, 0].astype(np.int64), g[case + "/hist0"][it])
    print("[idt %s] iterations >= 1: %d of %d bin indices on an edge (differ by one bin from the reference's)" % (
        case, n_edge, 3 * (rot.shape[0] - 1) * want_idx.shape[-1] if rot.shape[0] > 1 else 0))


@pytest.mark.parametrize("case,bins,n_iter", [("f64", 255, 4), ("f32", 255, 4), ("odd", 64, 2)])
def test_idt_small_bitwise_and_vs_reference(golden_dir, hip, case, bins, n_iter):
    g = _g(golden_dir, "idt_small.npz")
    t, r, rot = g[case + "/target"], g[case + "/reference"], g[case + "/rot"]
    out, d = run_hip(hip, t, r, rot, bins)
    assert_bitwise_vs_oracle(out, d, t, r, rot, bins)
    # against the real reference: iteration-0 integers exact, output within 1e-9
    assert np.array_equal(d["binidx"][0, 0].astype(np.uint16), g[case + "/binidx"][0])
    assert np.array_equal(d["hist"][0, 0, 0].astype(np.int64), g[case + "/hist0"][0])
    assert np.array_equal(d["hist"][0, 0, 1].astype(np.int64), g[case + "/hist1"][0])
    assert np.array_equal(d["par"][0, 0, :, 0:2], g[case + "/lohi"][0])
    assert_later_iterations_vs_reference(d, g, case, t, r, rot, bins)
    np.testing.assert_allclose(out, g[case + "/out"], rtol=0, atol=1e-9)


def test_idt_u8_256_vs_reference(golden_dir, hip):
    g = _g(golden_dir, "idt_u8_256.npz")
    t = g["target_u8"].astype(np.float32) / 255
    r = g["reference_u8"].astype(np.float32) / 255
    out, d = run_hip(hip, t, r, g["rot"])
    assert_bitwise_vs_oracle(out, d, t, r, g["rot"], 255)
    assert np.array_equal(d["binidx"][0, 0].astype(np.uint8), g["binidx_it0"])
    assert np.array_equal(d["hist"][0, 0, 0].astype(np.int64), g["hist0"][0])
    np.testing.assert_allclose(out[::3, ::3], g["out_s3"], rtol=0, atol=1e-9)


def test_idt_1080p_bitwise_vs_oracle_and_reference_samples(golden_dir, hip):
    g = _g(golden_dir, "idt_1080p.npz")
    rng = np.random.default_rng(int(g["seed"]))
    t = rng.random((1080, 1920, 3), dtype=np.float32)
    r = rng.random((1080, 1920, 3), dtype=np.float32)
    out, d = run_hip(hip, t, r, g["rot"])
    assert_bitwise_vs_oracle(out, d, t, r, g["rot"], 255)
    idx = g["idx"]
    assert np.array_equal(d["hist"][0, 0, 0].astype(np.int64), g["hist0"][0])
    assert np.array_equal(d["hist"][0, 0, 1].astype(np.int64), g["hist1"][0])
    assert np.array_equal(d["binidx"][0, 0][:, idx].astype(np.uint8), g["binidx_it0_samples"])
    np.testing.assert_allclose(out.reshape(-1, 3)[idx], g["out_samples"], rtol=0, atol=1e-9)
    # size-independent properties: every histogram sums to N; counts are non-negative
    assert (d["hist"].astype(np.int64).sum(axis=-1) == 1080 * 1920).all()


def test_idt_api_draws_rotations_like_reference(golden_dir, it):
    """Seeding numpy's global RNG reproduces the reference's matrices and therefore its output."""
    g = _g(golden_dir, "idt_small.npz")
    t, r = g["f64/target"], g["f64/reference"]
    np.random.seed(3)                      # the seed make_golden_idt.py used for this case
    out = it.iterative_distribution_transfer(t, r)
    assert out.dtype == np.float64 and out.shape == t.shape
    np.testing.assert_allclose(out, g["f64/out"], rtol=0, atol=1e-9)
    np.random.seed(4)
    out32 = it.iterative_distribution_transfer(g["f32/target"], g["f32/reference"])
    assert out32.dtype == np.float64
    np.testing.assert_allclose(out32, g["f32/out"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (2, 3), (5, 7), (33, 31)])
def test_idt_ragged_sizes_bitwise(hip, shape):
    rng = np.random.default_rng(shape[0] * 37 + shape[1])
    t = rng.random(shape + (3,))
    r = rng.random((4, 6, 3))
    rot = oit.draw_rotations(3)
    out, d = run_hip(hip, t, r, rot, 32)
    assert_bitwise_vs_oracle(out, d, t, r, rot, 32)


def test_idt_degenerate_constant_images(hip):
    """lo == hi on an axis: numpy widens the range by +-0.5 (_get_outer_edges)."""
    t = np.full((4, 4, 3), 0.25)
    r = np.full((3, 3, 3), 0.25)
    rot = np.eye(3)[None]
    out, d = run_hip(hip, t, r, rot, 16)
    assert_bitwise_vs_oracle(out, d, t, r, rot, 16)


def test_idt_batch_of_pairs(hip):
    rng = np.random.default_rng(8)
    t = rng.random((3, 9, 11, 3), dtype=np.float32)
    r = rng.random((3, 9, 11, 3), dtype=np.float32)
    rot = np.stack([oit.draw_rotations(2) for _ in range(3)])
    out = hip.idt(dev(t), dev(r), rot, bins=40).cpu().numpy()
    for b in range(3):
        ref = oit.iterative_distribution_transfer(t[b], r[b], bins=40, n_iter=2, rotations=rot[b])
        assert np.array_equal(out[b], ref)


def test_idt_n_iter_zero_and_empty(it):
    t = np.random.default_rng(0).random((3, 3, 3)).astype(np.float32)
    out = it.iterative_distribution_transfer(t, t, n_iter=0)
    assert out is t or np.array_equal(out, t)
    with pytest.raises(ValueError):
        it.iterative_distribution_transfer(t, np.zeros((0, 3, 3), np.float32))
