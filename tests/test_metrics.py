"""Per-frame quality metrics (Runner.test_step, reference methods/__init__.py:29-40).

not-gpu: the oracle (oracle/metrics.py) against its anchors -- scikit-image's structural_similarity for the SSIM core
(tests/golden/ssim_anchor.npz) and a run of the reference's own utils/icid.py for iCID (tests/golden/icid.npz);
gpu: the HIP kernels (ct_frame_ssim_f32 / ct_frame_icid_f32 / ct_frame_psnr_f32, through the C ABI) against the oracle and
the fixtures, with and without downsampling, plus size-independent properties at 1080p."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import metrics as om     # noqa: E402


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _u8(a):
    return torch.from_numpy(a.astype(np.float32) / np.float32(255))


def test_oracle_ssim_vs_skimage_anchor(golden_dir):
    g = _g(golden_dir, "ssim_anchor.npz")
    for tag in ("a", "b", "c"):
        x, y = torch.from_numpy(g[tag + "/x"])[None], torch.from_numpy(g[tag + "/y"])[None]
        assert abs(float(om.ssim(x, y)[0]) - float(g[tag + "/ssim"])) < 1e-9, tag
    x = torch.rand(2, 3, 40, 50)
    assert torch.allclose(om.ssim(x, x), torch.ones(2, dtype=torch.float64), atol=1e-12)
    assert om.metric_factor(1080, 1920) == 4 and om.metric_factor(383, 999) == 1 and om.metric_factor(384, 999) == 2
    assert om.metric_factor(640, 640) == 2 and om.metric_factor(650, 700) == 3      # round half to even: 2.5 -> 2


def test_oracle_icid_vs_reference_run(golden_dir):
    g = _g(golden_dir, "icid.npz")
    for tag in ("a", "b", "c", "d"):
        x, y = _u8(g[tag + "/x_u8"]), _u8(g[tag + "/y_u8"])
        got = float(om.icid(x, y))
        assert abs(got - float(g[tag + "/icid_f64"])) < 1e-10, tag           # float64 run of the reference
        assert abs(got - float(g[tag + "/icid_f32"])) < 5e-6, tag            # the reference as it runs (float32)
        assert float(om.icid(x, x)) == 0.0 == float(g[tag + "/icid_same"])


def test_oracle_fsim_properties():
    """piq.fsim restated (parity unpinned): structural properties of the restatement itself"""
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 48, 64, generator=gen)
    y = (x + 0.1 * torch.randn(2, 3, 48, 64, generator=gen)).clamp(0, 1)
    assert torch.allclose(om.fsim(x, x), torch.ones(2, dtype=torch.float64), atol=1e-12)
    a, b = om.fsim(x, y), om.fsim(y, x)
    assert torch.allclose(a, b, atol=1e-12) and (a < 1).all() and (a > 0.5).all()               # symmetric, degraded
    more = (x + 0.3 * torch.randn(2, 3, 48, 64, generator=gen)).clamp(0, 1)
    assert (om.fsim(x, more) < a).all()                                                           # monotone in the noise level
    f = om.fsim_filters(30, 45)
    assert f.shape == (16, 30, 45) and float(f[:, 0, 0].abs().max()) == 0.0 and float(f.min()) >= 0.0
    # the bank is symmetric under the half turn of the frequency plane for orientation 0 up to the angular spread: DC-free
    assert float(f.sum()) > 0


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


def _pair(seed, b, h, w, noise=0.05):
    gen = torch.Generator().manual_seed(seed)
    base = torch.rand(b, 3, h // 8 + 2, w // 8 + 2, generator=gen)
    gt = torch.nn.functional.interpolate(base, size=(h, w), mode="bicubic", align_corners=True).clamp(0, 1)
    gt = (gt + 0.04 * torch.rand(b, 3, h, w, generator=gen)).clamp(0, 1)
    x = (gt ** 1.15 * 0.93 + 0.02 + noise * torch.randn(b, 3, h, w, generator=gen)).clamp(0, 1)
    return x.contiguous(), gt.contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 56), (1, 11, 11), (1, 97, 130), (3, 400, 520), (1, 650, 700), (1, 1080, 1920), (1, 1083, 1925)])
def test_hip_ssim_vs_oracle(hip, shape):
    b, h, w = shape
    x, y = _pair(h + w, b, h, w)
    got = hip.frame_ssim(x.cuda(), y.cuda()).cpu()
    want = om.ssim(x, y)
    assert got.shape == (b,) and torch.allclose(got, want, rtol=0, atol=2e-6), (got, want)
    assert torch.allclose(hip.frame_ssim(x.cuda(), x.cuda()).cpu(), torch.ones(b, dtype=torch.float64), atol=1e-6)
    assert torch.equal(got, hip.frame_ssim(x.cuda(), y.cuda()).cpu())          # deterministic


@pytest.mark.gpu
def test_hip_ssim_vs_skimage_anchor(hip, golden_dir):
    g = _g(golden_dir, "ssim_anchor.npz")
    for tag in ("a", "b", "c"):
        x, y = torch.from_numpy(g[tag + "/x"])[None].cuda(), torch.from_numpy(g[tag + "/y"])[None].cuda()
        assert abs(float(hip.frame_ssim(x, y)[0]) - float(g[tag + "/ssim"])) < 2e-6, tag


@pytest.mark.gpu
def test_hip_icid_vs_reference_run(hip, golden_dir):
    g = _g(golden_dir, "icid.npz")
    for tag in ("a", "b", "c", "d"):
        x, y = _u8(g[tag + "/x_u8"]).cuda(), _u8(g[tag + "/y_u8"]).cuda()
        got = float(hip.frame_icid(x, y)[0])
        assert abs(got - float(g[tag + "/icid_f64"])) < 5e-6, (tag, got, float(g[tag + "/icid_f64"]))
        assert abs(float(hip.frame_icid(x, x)[0])) < 1e-6            # float32 maps: products of ones within rounding


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 6, 6), (3, 270, 480), (1, 1080, 1920), (1, 1083, 1925)])
def test_hip_icid_vs_oracle(hip, shape):
    b, h, w = shape
    x, y = _pair(h * 3 + w, b, h, w)
    got = hip.frame_icid(x.cuda(), y.cuda()).cpu()
    want = torch.stack([om.icid(x[i:i + 1], y[i:i + 1]) for i in range(b)])
    assert torch.allclose(got, want, rtol=0, atol=5e-6), (got, want)
    assert torch.equal(got, hip.frame_icid(x.cuda(), y.cuda()).cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 56), (1, 97, 131), (3, 270, 480), (1, 650, 700), (1, 1080, 1920), (2, 1083, 1925)])
def test_hip_fsim_vs_oracle(hip, shape):
    b, h, w = shape
    x, y = _pair(2 * h + w, b, h, w)
    got = hip.frame_fsim(x.cuda(), y.cuda()).cpu()
    want = om.fsim(x, y)
    print("\n[fsim %s] device %s oracle %s" % (shape, got.tolist(), want.tolist()))
    assert got.shape == (b,) and torch.allclose(got, want, rtol=0, atol=2e-5), (got, want)
    assert torch.allclose(hip.frame_fsim(x.cuda(), x.cuda()).cpu(), torch.ones(b, dtype=torch.float64), atol=1e-6)
    assert torch.equal(got, hip.frame_fsim(x.cuda(), y.cuda()).cpu())          # deterministic
    # batch composition does not matter (the median / thresholds are per image)
    if b > 1:
        assert torch.equal(got[1:2], hip.frame_fsim(x[1:2].cuda(), y[1:2].cuda()).cpu())


@pytest.mark.gpu
def test_hip_fsim_filter_bank_and_constants(hip):
    """the device-built log-Gabor bank and noise constants against the oracle's (float64)"""
    import ctypes
    h, w = 131, 200
    x = torch.rand(1, 3, h, w).cuda()
    hip.frame_fsim(x, x)
    filt, consts = hip._fsim_tables[(str(x.device), h, w)]
    want = om.fsim_filters(h, w).reshape(16, -1)
    assert float((filt.cpu().double() - want).abs().max()) < 1e-6
    fi = torch.fft.ifft2(om.fsim_filters(h, w)).real * (h * w) ** 0.5
    fi = fi.view(4, 4, h, w)
    em = (om.fsim_filters(h, w).view(4, 4, h, w)[:, 0] ** 2).sum((1, 2))
    an2 = (fi ** 2).sum((1, 2, 3))
    aij = sum((fi[:, s] * fi[:, t]).sum((1, 2)) for s in range(3) for t in range(s + 1, 4))
    got = consts.cpu()
    assert torch.allclose(got[:, 0], em, rtol=1e-5) and torch.allclose(got[:, 1], an2, rtol=1e-4) and torch.allclose(got[:, 2], aij, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_runner_test_step_reports_all_metrics(hip):
    from methods import METRICS, Runner
    x, y = _pair(9, 2, 64, 96)
    r = Runner("methods.linear.color_transfer_between_images")
    m = r.test_step({"target": x.cuda(), "reference": y.flip(3).contiguous().cuda(), "gt": y.cuda()})
    assert tuple(m.keys()) == METRICS and all(v.shape == (2,) and torch.isfinite(v).all() for v in m.values())
    res = r({"target": x.cuda(), "reference": y.flip(3).contiguous().cuda()}).clamp(0, 1).cpu()
    assert torch.allclose(m["Test PSNR"].cpu(), om.psnr(res, y), atol=1e-9)
    assert torch.allclose(m["Test SSIM"].cpu(), om.ssim(res, y), atol=2e-6)
    assert abs(float(m["Test iCID"][1]) - float(om.icid(res[1:2], y[1:2]))) < 5e-6
    assert torch.allclose(m["Test FSIM"].cpu(), om.fsim(res, y), atol=2e-5)
    with pytest.raises(hip.CtHipError):
        hip.frame_ssim(torch.rand(1, 3, 8, 8).cuda(), torch.rand(1, 3, 8, 8).cuda())      # smaller than the 11x11 window


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["table", "exact"])
def test_fused_reinhard_psnr(hip, mode):
    """ct_reinhard_psnr_f32 == ct_reinhard_f32 followed by ct_frame_psnr_f32 (same result image bit for bit, same PSNR)"""
    gen = torch.Generator().manual_seed(4)
    for shape in ((3, 67, 91, 3), (2, 256, 300, 3), (1, 5, 7, 3)):
        t, r, g = (torch.rand(shape, generator=gen).cuda() for _ in range(3))
        hip.set_lab_mode(mode)
        try:
            out, ps = hip.reinhard_psnr(t, r, g)
            want = hip.reinhard(t, r)
            want_ps = hip.frame_psnr(want, g)
        finally:
            hip.set_lab_mode("table")
        assert torch.equal(out, want)
        assert torch.allclose(ps, want_ps, rtol=1e-6, atol=0), (ps, want_ps)          # float32 per-tile partial sums vs float64
        assert torch.allclose(ps[:, 1].cpu(), om.psnr(out.cpu(), g.cpu()), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(270, 480), (256, 256), (135, 240), (2, 2), (1, 7), (17, 31), (64, 96), (45, 30), (97, 101), (8, 1),
                                   (540, 960), (250, 77), (1024, 4096), (3, 2 * 3 * 5 * 7 * 11)])
def test_fft2d_vs_torch(hip, shape):
    """csrc/fft2d.hip (the hand-written transform inside FSIM) against torch.fft in float64: every radix path -- 4 / 2 / 3 / 5 in
    registers, other primes (7, 11, 31, 97, 101) as plain butterflies, prime lengths, single rows / columns -- forward and
    unnormalised inverse, batched; error relative to the spectrum's largest magnitude."""
    h, w = shape
    g = torch.Generator().manual_seed(h * 1000 + w)
    b = 3 if h * w < 300000 else 1
    x = torch.complex(torch.randn(b, h, w, generator=g), torch.randn(b, h, w, generator=g)).to(torch.complex64)
    for inverse in (False, True):
        got = hip.fft2d_(x.cuda().clone(), inverse=inverse).cpu().to(torch.complex128)
        want = torch.fft.ifft2(x.to(torch.complex128)) * (h * w) if inverse else torch.fft.fft2(x.to(torch.complex128))
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err <= 2e-6, (shape, inverse, err)
    # round trip: ifft2(fft2(x)) = h w x
    y = hip.fft2d_(hip.fft2d_(x.cuda().clone()), inverse=True).cpu() / (h * w)
    assert (y - x).abs().max().item() <= 5e-6 * x.abs().max().item()
    with pytest.raises(hip.CtHipError):
        hip.fft2d_(torch.zeros(1, 4, 5000, dtype=torch.complex64, device="cuda"))          # an axis beyond 4096 points
