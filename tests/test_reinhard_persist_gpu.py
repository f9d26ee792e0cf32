"""GPU parity of the one-launch Reinhard transfer (csrc/reinhard_persist.hip; reference methods/linear.py:8-42): against the
float64 CPU oracle, against the two-sweep kernels it replaces, the uint8 front door against the float32 entry fed u8 / 255,
and the properties the in-launch hand-off must keep (bitwise reproducible, independent of the number of pairs per call)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import lab as olab        # noqa: E402
from oracle import linear as olin     # noqa: E402

RGB_TOL = 6.0e-6        # float32 output of the table arithmetic (tests/test_linear_gpu.py: RGB_TOL["table"]); saturated colours
LAB_TIGHT = 5e-5        # gate 1e-4
STATS_TOL = 3e-6        # Lab statistics of the float32 sweeps (tests/test_linear_gpu.py: STATS_TOL["table"])
PAIR_TOL = 6e-6         # persistent launch against the two sweeps: the target's statistics come from the apply-grade forward transform here


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    ct_hip.set_lab_mode("table")
    return ct_hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def lab_err(a, b):
    return np.abs(olab.rgb2lab(np.asarray(a, np.float64)) - olab.rgb2lab(np.asarray(b, np.float64))).max()


def two_sweep(hip, t, r):
    """the kernels of linear.hip by their own entries (lab_moments_lut_kernel x 2 + reinhard_apply_lut_kernel)"""
    st, sr = hip.lab_stats(t), hip.lab_stats(r)
    return hip.reinhard_apply(t, st, sr), st, sr


def test_1080p_vs_oracle_every_pixel_and_two_sweep(hip):
    rng = np.random.default_rng(99)
    t = rng.random((1080, 1920, 3), dtype=np.float32)
    r = (rng.random((1080, 1920, 3), dtype=np.float32) * 0.6 + 0.2).astype(np.float32)
    g = rng.random((1080, 1920, 3), dtype=np.float32)
    td, rd, gd = dev(t), dev(r), dev(g)
    stats = torch.zeros((2, 8), dtype=torch.float64, device="cuda")
    out, psnr = hip.reinhard_persist(td, rd, gt=gd, stats_out=stats, verify=True)
    out = out.cpu().numpy()
    ref = olin.color_transfer_between_images(t, r)
    assert np.abs(out - ref).max() <= RGB_TOL
    assert lab_err(out, ref) <= LAB_TIGHT
    # statistics: float32 per pixel values, float32 moment sums folded into an exact integer total
    mt, sdt = olin.lab_stats(t)
    mr, sdr = olin.lab_stats(r)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[0, 0:3], mt, rtol=0, atol=STATS_TOL)
    np.testing.assert_allclose(s[0, 3:6], sdt, rtol=0, atol=STATS_TOL)
    np.testing.assert_allclose(s[1, 0:3], mr, rtol=0, atol=STATS_TOL)
    np.testing.assert_allclose(s[1, 3:6], sdr, rtol=0, atol=STATS_TOL)
    assert s[0, 6] == s[1, 6] == 1080 * 1920
    # PSNR of the float32 result against gt, as piq.psnr computes it (methods/__init__.py:32)
    mse = np.mean((out.astype(np.float64) - g.astype(np.float64)) ** 2)
    p = psnr.cpu().numpy()[0]
    assert abs(p[0] - mse) <= 1e-9 * mse and abs(p[1] - 10 * np.log10(1 / mse)) <= 1e-8
    # the two-sweep kernels it replaces: same table arithmetic, statistics from a float32 sweep
    old, _, _ = two_sweep(hip, td, rd)
    assert np.abs(out - old.cpu().numpy()).max() <= PAIR_TOL
    # the fused entries (two sweeps unless CT_HIP_REINHARD_PERSIST=1): same arithmetic per pixel, statistics from their own sums
    o2, p2 = hip.reinhard_psnr(td, rd, gd)
    assert (o2 - torch.from_numpy(out).cuda()).abs().max().item() <= PAIR_TOL and abs(p2[0, 1].item() - p[1]) <= 1e-6
    if hip.reinhard_takes_persist(1080 * 1920):
        assert torch.equal(o2, torch.from_numpy(out).cuda()) and torch.equal(p2, psnr)
    # run-to-run: bitwise
    out3, psnr3 = hip.reinhard_persist(td, rd, gt=gd)
    assert np.array_equal(out3.cpu().numpy(), out) and torch.equal(psnr3, psnr)


def test_batch_independence_and_pipeline(hip):
    """R(p+1) | wait(p) | A(p) | T(p+1): results of a pair must not depend on its neighbours or on the batch size."""
    rng = np.random.default_rng(7)
    B = 5
    t = rng.random((B, 1080, 1920, 3), dtype=np.float32)
    r = rng.random((B, 1080, 1920, 3), dtype=np.float32) * np.linspace(0.3, 1.0, B, dtype=np.float32)[:, None, None, None]
    g = rng.random((B, 1080, 1920, 3), dtype=np.float32)
    td, rd, gd = dev(t), dev(r), dev(g)
    out, psnr = hip.reinhard_persist(td, rd, gt=gd, verify=True)
    for b in (0, 2, 4):
        o1, p1 = hip.reinhard_persist(td[b:b + 1], rd[b:b + 1], gt=gd[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(p1[0], psnr[b])
    o_nogt = hip.reinhard_persist(td, rd)
    assert torch.equal(o_nogt, out)
    ref = olin.color_transfer_between_images(t[3], r[3])
    assert np.abs(out[3].cpu().numpy() - ref).max() <= RGB_TOL


@pytest.mark.parametrize("n_pixels", [256, 257, 511, 1000, 70001, 300 * 301])
def test_small_and_ragged_sizes(hip, n_pixels):
    """sizes below the automatic dispatch: idle waves, idle workgroups, the ragged rest of the last tile"""
    assert hip.reinhard_persist_supported(n_pixels)
    rng = np.random.default_rng(n_pixels)
    t = rng.random((2, n_pixels, 1, 3), dtype=np.float32)
    r = rng.random((2, n_pixels, 1, 3), dtype=np.float32) * 0.5 + 0.25
    r = r.astype(np.float32)
    g = rng.random((2, n_pixels, 1, 3), dtype=np.float32)
    out, psnr = hip.reinhard_persist(dev(t), dev(r), gt=dev(g), verify=True)
    out = out.cpu().numpy()
    for b in range(2):
        ref = olin.color_transfer_between_images(t[b], r[b])
        # few pixels: the float32 rounding of a moment term (~3e-8 x 500 in a*) is not averaged away
        assert np.abs(out[b] - ref).max() <= RGB_TOL, n_pixels
        mse = np.mean((out[b].astype(np.float64) - g[b].astype(np.float64)) ** 2)
        assert abs(psnr[b, 0].item() - mse) <= 2e-7 * mse        # one float32 fmaf chain of 12 squares per lane and tile, float64 from there


def test_size_limits_and_large_frames(hip):
    assert not hip.reinhard_persist_supported(255)
    with pytest.raises(hip.CtHipError):
        hip.reinhard_persist(torch.rand(1, 100, 1, 3, device="cuda"), torch.rand(1, 100, 1, 3, device="cuda"))
    # 3840 x 2160: most of a workgroup's tiles do not fit its LDS and are fetched a second time -- same results as the two sweeps
    assert hip.reinhard_persist_supported(3840 * 2160)
    t, r = torch.rand(1, 2160, 3840, 3, device="cuda"), torch.rand(1, 2160, 3840, 3, device="cuda") * 0.5
    old, _, _ = two_sweep(hip, t, r)
    assert (hip.reinhard_persist(t, r, verify=True) - old).abs().max().item() <= PAIR_TOL
    assert not hip.reinhard_takes_persist(3840 * 2160)         # the fused entry keeps the two sweeps at this size in any case
    assert (hip.reinhard(t, r) - old).abs().max().item() <= PAIR_TOL


def _special(h, w):
    rng = np.random.default_rng(5)
    u = rng.random((h, w, 3), dtype=np.float32)
    yield "u8-levels", (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255), (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], -1).astype(np.float32)
    yield "graded", (0.8 * g + 0.1 * u).astype(np.float32), (0.5 * g[::-1] + 0.3).astype(np.float32)
    yield "dark", (u * 0.12).astype(np.float32), (u[::-1] * 0.2).astype(np.float32)
    o = u.copy()
    o[::7, ::5] = 1.5
    o[::11, ::3] = -0.25
    yield "out-of-range", o, u
    o2 = u.copy()
    o2[100:103, 200:260] = 1.25                      # a few tiles only: exact-code tiles beside parked ones
    yield "out-of-range-sparse", o2, o2[::-1].copy()


@pytest.mark.parametrize("size", [(1080, 1920), (270, 483)])
def test_all_branches_vs_oracle(hip, size):
    for name, t, r in _special(*size):
        lt, lr = olab.rgb2lab(t.astype(np.float64)), olab.rgb2lab(r.astype(np.float64))
        mt, sdt = lt.reshape(-1, 3).mean(0), lt.reshape(-1, 3).std(0)
        mr, sdr = lr.reshape(-1, 3).mean(0), lr.reshape(-1, 3).std(0)
        rgb_ref = olab.lab2rgb((lt - mt) * (sdr / sdt) + mr)
        out = hip.reinhard_persist(dev(t[None]), dev(r[None]), verify=True)[0].cpu().numpy()
        assert out.min() >= 0 and out.max() <= 1
        assert np.abs(out - rgb_ref).max() <= RGB_TOL, name
        assert lab_err(out, rgb_ref) <= LAB_TIGHT, name


def test_nan_inf_huge_and_constant_frames(hip):
    """sums that do not fit the integer hand-off take the float64 slab; statistics and results then behave like the exact
    path of the two-sweep kernels (and like the reference: nan / inf propagate)."""
    rng = np.random.default_rng(1)
    u = rng.random((1, 400, 500, 3), dtype=np.float32)
    v = rng.random((1, 400, 500, 3), dtype=np.float32)
    # NaN pixel in the target: every statistic of the target is NaN, the whole result is NaN
    n = u.copy()
    n[0, 5, 7, 1] = np.nan
    st = torch.zeros((2, 8), dtype=torch.float64, device="cuda")
    out = hip.reinhard_persist(dev(n), dev(v), stats_out=st, verify=True).cpu().numpy()
    assert np.isnan(st.cpu().numpy()[0, :6]).all() and np.isfinite(st.cpu().numpy()[1, :6]).all()
    assert np.isnan(out).all()
    # a huge but finite value: float64 statistics like the exact code (the reference computes finite garbage here too)
    h = u.copy()
    h[0, 9, 9, 0] = 3e4
    hip.set_lab_mode("exact")
    st_exact = torch.stack([hip.lab_stats(dev(h))[0], hip.lab_stats(dev(v))[0]]).cpu().numpy()
    out_exact = hip.reinhard_apply(dev(h), hip.lab_stats(dev(h)), hip.lab_stats(dev(v))).cpu().numpy()
    hip.set_lab_mode("table")
    out = hip.reinhard_persist(dev(h), dev(v), stats_out=st, verify=True).cpu().numpy()
    np.testing.assert_allclose(st.cpu().numpy()[:, :6], st_exact[:, :6], rtol=1e-6)
    assert np.abs(out - out_exact).max() <= 1e-5
    # constant target: sigma_t = 0 -> inf / nan coefficients -> nothing finite (tests/test_linear_gpu.py, same expectation)
    c = np.full((1, 400, 500, 3), 0.3, np.float32)
    out = hip.reinhard_persist(dev(c), dev(u), verify=True).cpu().numpy()
    assert not np.isfinite(out).any()
    # and a healthy pair right after, in the same batch as a poisoned one
    tb = np.concatenate([n, u]); rb = np.concatenate([v, v])
    out = hip.reinhard_persist(dev(tb), dev(rb), verify=True).cpu().numpy()
    assert np.isnan(out[0]).all()
    assert np.abs(out[1] - olin.color_transfer_between_images(u[0], v[0])).max() <= RGB_TOL


def test_u8_front_door(golden_dir, hip):
    """ct_reinhard_psnr_u8 reads the bytes and takes k / 255 (IEEE float32 division, the reference's `.float() / 255`,
    utils/data.py:84) and its gamma expansion from 256-entry tables filled by the float32 kernel's own functions: per pixel the
    arithmetic is the float32 entry's on u8.float() / 255.  The frame statistics agree to their last float32 rounding only (the
    uint8 tile puts other pixels on a lane, so the moment terms are added in another order): results within PAIR_TOL, PSNR 1e-6 rel."""
    g = np.load(os.path.join(golden_dir, "linear_u8_256.npz"), allow_pickle=False)
    t8, r8 = g["target_u8"], g["reference_u8"]
    out = hip.reinhard_persist(dev(t8[None]), dev(r8[None]), verify=True)[0].cpu().numpy()
    sl = (slice(None, None, 3), slice(None, None, 3))
    np.testing.assert_allclose(out[sl], g["reinhard_s3"], rtol=0, atol=RGB_TOL)          # the reference's own result on this frame
    assert lab_err(out[sl], g["reinhard_s3"]) <= LAB_TIGHT
    # the reference's conversion as its dataloader does it -- on the CPU, an IEEE division (utils/data.py:84); torch's GPU kernel for
    # `tensor / scalar` multiplies by the reciprocal instead, which is off by one ulp on 126 of the 256 levels
    f = lambda a: (torch.from_numpy(a).float() / 255).cuda()
    assert np.abs(out - hip.reinhard_persist(f(t8[None]), f(r8[None]))[0].cpu().numpy()).max() <= PAIR_TOL   # the statistics' float32 sums run in another order; saturated colours amplify (tests/test_linear_gpu.py: RGB_TOL)
    # 1080p with the metric, odd tail included
    rng = np.random.default_rng(3)
    for shape in ((2, 1080, 1920, 3), (1, 1080 * 1920 + 131, 1, 3)):
        t, r, gt = (rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(3))
        st8, stf = (torch.zeros((2 * shape[0], 8), dtype=torch.float64, device="cuda") for _ in range(2))
        o8, p8 = hip.reinhard_persist(dev(t), dev(r), gt=dev(gt), stats_out=st8, verify=True)
        of, pf = hip.reinhard_persist(f(t), f(r), gt=f(gt), stats_out=stf, verify=True)
        assert (o8 - of).abs().max().item() <= PAIR_TOL
        assert ((p8 - pf).abs() <= 1e-6 * pf.abs()).all()
        assert (st8 - stf).abs().max().item() <= 1e-6
        ref = olin.color_transfer_between_images((t[0].astype(np.float32) / 255).reshape(-1, 1, 3), (r[0].astype(np.float32) / 255).reshape(-1, 1, 3))
        assert np.abs(o8[0].cpu().numpy().reshape(-1, 1, 3) - ref).max() <= RGB_TOL
    # run to run: bitwise
    o8b, p8b = hip.reinhard_persist(dev(t), dev(r), gt=dev(gt))
    assert torch.equal(o8, o8b) and torch.equal(p8, p8b)


def test_graph_capture_and_replay(hip):
    """memset + persistent kernel + PSNR finish are plain stream work: capturable in a hipGraph and replayable on new data
    (the ABI never allocates or synchronises; epochs / arrival counts live in the per-call records the memset node zeroes)."""
    rng = np.random.default_rng(21)
    mk = lambda: dev(rng.random((2, 1080, 1920, 3), dtype=np.float32))
    t, r, g = mk(), mk(), mk()
    out = torch.empty_like(t)
    ps = torch.zeros((2, 2), dtype=torch.float64, device="cuda")
    hip.reinhard_persist(t, r, gt=g, out=out, psnr_out=ps)            # warm: workspace allocation, function attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.reinhard_persist(t, r, gt=g, out=out, psnr_out=ps)
    for _ in range(2):
        t2, r2, g2 = mk(), mk(), mk()
        want_o, want_p = hip.reinhard_persist(t2, r2, gt=g2)
        want_o, want_p = want_o.clone(), want_p.clone()
        t.copy_(t2); r.copy_(r2); g.copy_(g2)
        out.zero_(); ps.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want_o) and torch.equal(ps, want_p)


def test_beside_another_streams_work(hip):
    """The grid needs every workgroup resident for its in-launch hand-off.  Other work on the GPU only delays that (its
    workgroups drain, ours take the CUs as they free up): results unchanged, error word 0, no give-up of a bounded spin."""
    rng = np.random.default_rng(22)
    t, r, g = (dev(rng.random((3, 1080, 1920, 3), dtype=np.float32)) for _ in range(3))
    want_o, want_p = hip.reinhard_persist(t, r, gt=g, verify=True)
    want_o, want_p = want_o.clone(), want_p.clone()
    a = torch.randn(8192, 8192, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(6):
                b = a @ a                                              # ~tens of ms of full-chip work on another stream
        got_o, got_p = hip.reinhard_persist(t, r, gt=g, verify=True)  # verify: synchronises and checks the error word
        assert torch.equal(got_o, want_o) and torch.equal(got_p, want_p)
    side.synchronize()
    del b


def test_more_pairs_than_one_launch_holds(hip):
    """batches beyond 64 pairs run as several launches (the pivots of 2 x 64 images live in LDS): same results as pair by pair, the
    caller's statistics records [targets of the whole batch][references of the whole batch] filled across the launches"""
    rng = np.random.default_rng(70)
    B, n = 70, 3000
    t = rng.random((B, n, 1, 3), dtype=np.float32)
    r = (rng.random((B, n, 1, 3), dtype=np.float32) * np.linspace(0.2, 1.0, B, dtype=np.float32)[:, None, None, None]).astype(np.float32)
    g = rng.random((B, n, 1, 3), dtype=np.float32)
    stats = torch.zeros((2 * B, 8), dtype=torch.float64, device="cuda")
    out, psnr = hip.reinhard_persist(dev(t), dev(r), gt=dev(g), stats_out=stats, verify=True)
    for b in (0, 63, 64, 69):
        s1 = torch.zeros((2, 8), dtype=torch.float64, device="cuda")
        o1, p1 = hip.reinhard_persist(dev(t[b:b + 1]), dev(r[b:b + 1]), gt=dev(g[b:b + 1]), stats_out=s1)
        assert torch.equal(o1[0], out[b]) and torch.equal(p1[0], psnr[b])
        assert torch.equal(s1[0], stats[b]) and torch.equal(s1[1], stats[B + b])
        ref = olin.color_transfer_between_images(t[b], r[b])
        assert np.abs(out[b].cpu().numpy() - ref).max() <= RGB_TOL


def test_two_streams_are_chained_and_status_is_clean(hip):
    """Two persistent launches must not share the GPU (each needs all of its workgroups resident): launches of one device are
    chained through an event, so calls from two streams run one after the other instead of starving each other until the
    bounded spins give up.  The device's sticky status stays 0 (include/ct_hip.h: ct_device_status)."""
    import time
    rng = np.random.default_rng(31)
    t, r, g = (dev(rng.random((4, 1080, 1920, 3), dtype=np.float32)) for _ in range(3))
    want_o, want_p = hip.reinhard_persist(t, r, gt=g, verify=True)
    want_o, want_p = want_o.clone(), want_p.clone()
    assert hip.device_status(clear=True) == 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        for s in (sa, sb):
            with torch.cuda.stream(s):
                outs.append(hip.reinhard_persist(t, r, gt=g))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0                       # a starved launch would sit in 2 s spins
    for o, p in outs:
        assert torch.equal(o, want_o) and torch.equal(p, want_p)
    assert hip.device_status() == 0


def test_two_streams_of_different_element_types_are_chained(hip):
    """the launch chain is ONE per device, not one per element type -- a float32 launch on one stream and a uint8 launch
    on another would starve each other exactly like two launches of one type.  Alternating types on two streams: every result is
    that of the call run alone, nothing sits in a 2 s spin, the sticky status stays 0."""
    import time
    rng = np.random.default_rng(32)
    t, r, g = (dev(rng.random((4, 1080, 1920, 3), dtype=np.float32)) for _ in range(3))
    t8, r8, g8 = ((x * 255).round().to(torch.uint8) for x in (t, r, g))
    want = [tuple(x.clone() for x in hip.reinhard_persist(a, b, gt=c, verify=True)) for a, b, c in ((t, r, g), (t8, r8, g8))]
    assert hip.device_status(clear=True) == 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        for k, s in enumerate((sa, sb)):
            kind = (i + k) & 1                                  # both orders: f32 after u8 and u8 after f32, on either stream
            with torch.cuda.stream(s):
                outs.append((kind, hip.reinhard_persist(*((t, r), (t8, r8))[kind], gt=(g, g8)[kind])))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0
    for kind, (o, p) in outs:
        assert torch.equal(o, want[kind][0]) and torch.equal(p, want[kind][1])
    assert hip.device_status() == 0


def test_output_must_not_overlap_an_input(hip):
    """include/ct_hip.h: a flagged tile is redone from the INPUT frame after its fast-path result has been stored, so the persistent
    launch refuses an output that overlaps target / reference / gt (the two-sweep entries allow out == target)"""
    rng = np.random.default_rng(33)
    t, r = dev(rng.random((1, 64, 64, 3), dtype=np.float32)), dev(rng.random((1, 64, 64, 3), dtype=np.float32))
    with pytest.raises(hip.CtHipError):
        hip.reinhard_persist(t, r, out=t)
    with pytest.raises(hip.CtHipError):
        hip.reinhard_persist(t, r, out=r)
    buf = torch.empty(2 * t.numel(), dtype=torch.float32, device="cuda")
    buf[: t.numel()] = t.flatten()
    with pytest.raises(hip.CtHipError):                          # a partial overlap
        hip.reinhard_persist(buf[: t.numel()].view(t.shape), r, out=buf[t.numel() // 2: t.numel() // 2 + t.numel()].view(t.shape))
    hip.reinhard_persist(t, r, verify=True)                      # and the plain call still runs
