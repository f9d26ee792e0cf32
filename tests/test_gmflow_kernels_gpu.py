"""Unit parity of every GMFlow building-block kernel (csrc/gmflow.hip) against the torch op(s) of the reference it
replaces, computed on the CPU in float64 where that is meaningful.  Tolerances are float32 rounding level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F   # noqa: E402

from oracle import gmflow as og   # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


def close(a, b, msg, atol=2e-5, rtol=2e-5):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


G = torch.Generator().manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, generator=G)


@pytest.mark.parametrize("cfg", [  # (n, cin, cout, h, w, kh, kw, stride, ph, pw)
    (2, 3, 64, 37, 70, 7, 7, 2, 3, 3), (1, 64, 96, 19, 35, 3, 3, 2, 1, 1), (1, 64, 96, 19, 35, 1, 1, 2, 0, 0),
    (1, 128, 128, 12, 33, 3, 3, 1, 1, 1), (1, 128, 128, 13, 34, 3, 3, 2, 1, 1), (2, 384, 128, 9, 40, 1, 5, 1, 0, 2),
    (1, 384, 128, 10, 33, 5, 1, 1, 2, 0), (1, 2, 128, 11, 37, 7, 7, 1, 3, 3), (1, 81, 256, 8, 32, 1, 1, 1, 0, 0),
    (1, 256, 126, 8, 36, 3, 3, 1, 1, 1), (1, 256, 2, 8, 36, 3, 3, 1, 1, 1), (1, 256, 144, 8, 36, 1, 1, 1, 0, 0),
    # multi-tile / multi-group geometries of the LDS-tiled stride-1 path (persistent schedule, 16-byte I/O)
    (2, 96, 96, 40, 72, 3, 3, 1, 1, 1), (1, 130, 256, 24, 64, 3, 3, 1, 1, 1), (3, 384, 128, 20, 68, 1, 5, 1, 0, 2),
    (3, 384, 128, 21, 68, 5, 1, 1, 2, 0), (2, 64, 192, 17, 100, 1, 1, 1, 0, 0),
    # the direct VALU kernels (csrc/conv_direct.hip): cout <= 4, and cin <= 3 with a 7x7 kernel; several tiles, ragged edges
    (2, 256, 2, 37, 70, 3, 3, 1, 1, 1), (1, 61, 3, 20, 45, 3, 3, 1, 1, 1), (1, 16, 4, 9, 33, 1, 1, 1, 0, 0),
    (2, 2, 128, 40, 72, 7, 7, 1, 3, 3), (1, 3, 64, 64, 131, 7, 7, 2, 3, 3),
    # stride 2 on the MFMA tile kernel through the space-to-depth image (even height, width % 8 == 0; split mode)
    (2, 64, 96, 40, 72, 3, 3, 2, 1, 1), (2, 64, 96, 40, 72, 1, 1, 2, 0, 0), (1, 96, 128, 24, 64, 3, 3, 2, 1, 1),
    (2, 128, 128, 16, 56, 3, 3, 2, 1, 1), (1, 40, 70, 18, 40, 3, 3, 2, 1, 1)])
def test_gconv(hip, cfg, conv_mode):
    n, cin, cout, h, w, kh, kw, s, ph, pw = cfg
    x, wt, b = rnd(n, cin, h, w), rnd(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5, rnd(cout)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=s, padding=(ph, pw))
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    for act, fn in ((0, lambda t: t), (2, torch.relu), (3, torch.sigmoid), (4, torch.tanh)):
        out = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), s, (ph, pw), act=act)
        close(out, fn(ref), "gconv %s act %d" % (cfg, act))
    wp, _ = hip.pack_gconv_weight(wt.cuda(), None)
    close(hip.gconv2d(x.cuda(), wp, None, cout, (kh, kw), s, (ph, pw)), ref - b.double().view(1, -1, 1, 1), "no bias")


@pytest.mark.parametrize("cfg", [(1, 16, 2, 12, 40, 3), (1, 8, 4, 9, 33, 1), (2, 32, 3, 20, 45, 3)])
def test_small_cout_conv_zero_padding_is_a_select(hip, cfg):
    """csrc/conv_direct.hip pads by clamping the tap address to pixel (0, 0): a non-finite value there must stay local (real zero
    padding, the reference's F.conv2d, the MFMA kernels), not turn every border pixel into NaN through inf * 0"""
    n, cin, cout, h, w, k = cfg
    x, wt, b = rnd(n, cin, h, w), rnd(cout, cin, k, k) / (cin * k * k) ** 0.5, rnd(cout)
    x[:, 0, 0, 0] = float("inf")
    x[:, 1, 0, 0] = float("nan")
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=k // 2)
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    out = hip.gconv2d(x.cuda(), wp, bp, cout, (k, k), 1, (k // 2, k // 2)).cpu()
    bad_ref = ~torch.isfinite(ref)
    assert bad_ref.sum() <= n * cout * k * k                       # only the taps that really see pixel (0, 0)
    assert torch.equal(~torch.isfinite(out), bad_ref)
    close(out[~bad_ref], ref[~bad_ref], "small-cout conv beside a non-finite corner")


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 24, 64, 3, 3), (1, 3, 64, 16, 32, 3, 3), (1, 64, 32, 9, 36, 3, 3), (1, 32, 3, 16, 40, 3, 3),
    (1, 129, 64, 8, 32, 1, 1), (2, 384, 128, 20, 68, 1, 5), (2, 384, 128, 21, 68, 5, 1), (1, 130, 256, 24, 64, 3, 3),
    (1, 81, 256, 8, 32, 1, 1)])
def test_conv_split_bf16(hip, cfg):
    """three-way bf16 split of both operands, six MFMAs per product: float32-grade accuracy (csrc/conv_split.hip)"""
    n, cin, cout, h, w, kh, kw = cfg
    x, wt, b = rnd(n, cin, h, w) * 3, rnd(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5, rnd(cout)
    res = rnd(n, cout, h, w)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=(kh // 2, kw // 2))
    ws, b64 = hip.pack_conv_weight_split(wt.cuda(), b.cuda())
    out = torch.empty(n, cout, h, w, device="cuda")
    tol = 2e-6 * max(1.0, ref.abs().max().item())          # float32 accumulation of ~cin*kh*kw terms of this magnitude
    for act, fn in ((0, lambda t: t), (1, lambda t: F.leaky_relu(t, 0.01)), (2, torch.relu), (3, torch.sigmoid), (4, torch.tanh)):
        hip._conv_split(x.cuda(), (ws, b64), cout, kh, kw, act, None, False, out)
        err = (out.double().cpu() - fn(ref)).abs().max().item()
        assert err < tol, ("split conv", cfg, act, err)
    hip._conv_split(x.cuda(), (ws, b64), cout, kh, kw, 0, res.cuda(), True, out)          # ResB skip in the accumulators + clamp
    close(out, (ref + res.double()).clamp(0, 1), "split conv + skip + clamp", atol=tol, rtol=0)
    hip._conv_split(x.cuda(), (ws, b64), cout, kh, kw, 2, res.cuda(), False, out)         # activation then skip
    close(out, torch.relu(ref) + res.double(), "split conv, relu, + skip", atol=tol, rtol=0)
    if cin >= 48:                                  # two-source input == conv on torch.cat([a, b], 1)
        c1 = 16 * (cin // 32)
        xa, xb = x[:, :c1].contiguous().cuda(), x[:, c1:].contiguous().cuda()
        hip._conv_split(xa, (ws, b64), cout, kh, kw, 2, None, False, out, x2=xb)
        close(out, torch.relu(ref), "split conv on a two-source input", atol=tol, rtol=0)
        wpg, bpg = hip.pack_gconv_weight(wt.cuda(), b.cuda())
        close(hip.gconv2d(xa, wpg, bpg, cout, (kh, kw), 1, (kh // 2, kw // 2), act=2, x2=xb), torch.relu(ref), "gconv2d x2", atol=tol, rtol=0)
    # scaling the input by a power of two scales every bf16 piece and every partial sum exactly: bitwise equivariance
    ws0, z64 = hip.pack_conv_weight_split(wt.cuda(), None)
    o1, o2 = torch.empty_like(out), torch.empty_like(out)
    hip._conv_split(x.cuda(), (ws0, z64), cout, kh, kw, 0, None, False, o1)
    hip._conv_split((x * 1024.0).cuda(), (ws0, z64), cout, kh, kw, 0, None, False, o2)
    assert torch.equal(o1 * 1024.0, o2), "split conv is not exactly equivariant under power-of-two scaling"
    # the exact-f32 kernel on the same operands: the two agree to float32 rounding level
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    hip.set_conv_mode("exact")
    try:
        exact = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, (kh // 2, kw // 2))
    finally:
        hip.set_conv_mode("split")
    hip._conv_split(x.cuda(), (ws, b64), cout, kh, kw, 0, None, False, out)
    assert (out - exact).abs().max().item() < 2 * tol


def test_instance_norm(hip):
    x, skip = rnd(2, 5, 17, 23) * 3 + 1, rnd(2, 5, 17, 23)
    ref = F.instance_norm(x.double(), eps=1e-5)
    close(hip.instance_norm(x.cuda(), 0), ref, "IN")
    close(hip.instance_norm(x.cuda(), 1), torch.relu(ref), "IN+relu")
    close(hip.instance_norm(x.cuda(), 2, skip.cuda()), torch.relu(skip.double() + torch.relu(ref)), "IN+relu+skip+relu")
    # few large planes: every plane is split over 16 workgroups (partial sums + apply); ragged chunking (plane % 64 != 0)
    x, skip = rnd(1, 3, 131, 127) * 2 - 0.5, rnd(1, 3, 131, 127)
    ref = F.instance_norm(x.double(), eps=1e-5)
    close(hip.instance_norm(x.cuda(), 0), ref, "IN split")
    close(hip.instance_norm(x.cuda(), 2, skip.cuda()), torch.relu(skip.double() + torch.relu(ref)), "IN+relu+skip+relu split")


def test_eltwise(hip):
    a, b, c = rnd(2, 6, 5, 7), rnd(2, 6, 5, 7), rnd(2, 6, 5, 7)
    close(hip.eltwise(0, a.cuda(), b.cuda()), a + b, "add")
    close(hip.eltwise(1, a.cuda(), b.cuda()), a * b, "mul")
    z = torch.sigmoid(a)
    close(hip.eltwise(2, z.cuda(), b.cuda(), c.cuda()), (1 - z) * b + z * c, "gru")
    img = torch.rand(2, 3, 5, 7, generator=G) * 255
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    close(hip.eltwise(3, img.cuda(), plane=35), (img / 255.0 - mean) / std, "normalize_img", atol=1e-5)
    close(hip.eltwise(4, a.cuda(), s0=2.5), a * 2.5, "scale")
    close(hip.eltwise(5, a.cuda(), plane=35, chans=6, split=3), torch.cat([torch.tanh(a[:, :3]), torch.relu(a[:, 3:])], 1), "tanh|relu")


@pytest.mark.parametrize("mode", ["split", "exact"])
@pytest.mark.parametrize("t,k,n,act", [(200, 128, 128, 0), (1000, 256, 1024, 6), (77, 1024, 128, 0), (33, 128, 128, 0), (300, 160, 200, 6),
                                       (129, 32, 2, 0), (4000, 128, 384, 0), (1000, 128, 96, 6), (70000, 128, 128, 0), (31, 128, 3, 0)])
def test_linear_tokens(hip, t, k, n, act, mode):
    """both arithmetic paths of nn.Linear on tokens (float32 MFMA / three-piece bf16 split, six MFMAs per product) against
    float64 torch -- the split path is held to the same float32-rounding-level bound"""
    x, w, b = rnd(t, k), rnd(n, k) / k ** 0.5, rnd(n)
    ref = F.linear(x.double(), w.double(), b.double())
    if act == 6:
        ref = F.gelu(ref)
    wc = w.cuda()
    close(hip.linear_tokens(x.cuda(), wc, b.cuda(), act=act, mode=mode), ref, "linear")
    close(hip.linear_tokens(x.cuda(), wc, None, act=0, mode=mode), F.linear(x.double(), w.double()), "linear nobias")
    if k >= 64:                                    # [x[:, :k1] | x[:, k1:]] from two tensors == linear on the concatenation
        k1 = 32 * (k // 64)
        xa, xb = x[:, :k1].contiguous().cuda(), x[:, k1:].contiguous().cuda()
        close(hip.linear_tokens(xa, wc, b.cuda(), act=act, x2=xb, mode=mode), ref, "linear on a two-source row")
    if mode == "split":                            # the pre-split weight is cached on the tensor and follows in-place updates
        assert wc._ct_lin_split[1].shape == ((n + 127) // 128, k // 32, 3, 4, 128, 8)
        wc.mul_(2.0)
        close(hip.linear_tokens(x.cuda(), wc, None, mode=mode), 2 * F.linear(x.double(), w.double()), "repacked after an in-place update")


def test_linear_tokens_split_wide_dynamic_range(hip):
    """split-path accuracy does not depend on operand magnitude: rows scaled by 2^-20 .. 2^20 keep the relative error of a
    float32 dot product (the three bf16 pieces carry 24 mantissa bits at any exponent)"""
    t, k, n = 512, 256, 128
    x, w = rnd(t, k), rnd(n, k) / 16
    scale = torch.pow(2.0, torch.randint(-20, 21, (t, 1), generator=G).float())
    xs = x * scale
    ref = F.linear(xs.double(), w.double())
    out = hip.linear_tokens(xs.cuda(), w.cuda(), None, mode="split").cpu().double()
    bound = F.linear(xs.double().abs(), w.double().abs())            # sum |x||w|: the natural error scale of a dot product
    assert ((out - ref).abs() / bound).max().item() < 4e-7


def test_attention_kv_shift_is_the_swapped_concat(hip):
    """cross attention against the other half of the batch (transformer.py:281-287's concat1) without the copy: reading keys /
    values half the rows further == attending to torch.cat(chunk(2)[::-1]) -- bit for bit (same arithmetic, same order)"""
    b, l = 4, 224
    q, k, v = rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, 128)
    perm = torch.stack([torch.randperm(l, generator=G) + i * l for i in range(b)])          # a window-style row table
    rowmap = perm.to(torch.int32).cuda()
    region = torch.randint(0, 3, (b, l), generator=G, dtype=torch.int32).cuda()
    swap = lambda t: torch.cat(t.chunk(2, dim=0)[::-1], dim=0).contiguous()
    want = hip.attention_tokens(q.cuda(), swap(k).cuda(), swap(v).cuda(), region, rowmap=rowmap)
    got = hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), region, rowmap=rowmap, kv_shift=b * l // 2)
    assert torch.equal(got, want)
    with pytest.raises(hip.CtHipError):
        hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), kv_shift=5)                        # only with a row table


def test_layernorm(hip):
    x, g, b, r = rnd(3, 50, 128) * 2 + 0.5, rnd(128), rnd(128), rnd(3, 50, 128)
    ref = F.layer_norm(x.double(), (128,), g.double(), b.double())
    close(hip.layernorm128(x.cuda(), g.cuda(), b.cuda()), ref, "LN")
    close(hip.layernorm128(x.cuda(), g.cuda(), b.cuda(), r.cuda()), r.double() + ref, "res+LN")


@pytest.mark.parametrize("b,l,cv,masked", [(2, 160, 128, False), (3, 448, 128, True), (1, 1792, 128, True), (2, 700, 2, False)])
def test_attention_tokens(hip, b, l, cv, masked):
    q, k, v = rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, cv)
    scores = torch.matmul(q.double(), k.double().transpose(1, 2)) / 128 ** 0.5
    region = None
    if masked:
        region = torch.randint(0, 4, (b, l), generator=G, dtype=torch.int32)
        scores = scores + torch.where(region[:, :, None] != region[:, None, :], -100.0, 0.0)
    ref = torch.matmul(torch.softmax(scores, dim=-1), v.double())
    for nsplit in (None, 1, 3, 8):                 # key split + merge kernel: same softmax
        out = hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), region.cuda() if masked else None, nsplit=nsplit)
        close(out, ref, "attention nsplit=%s" % nsplit, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("cv,shift", [(128, False), (128, True), (2, True)])
def test_attention_rowmap_is_window_partition(hip, cv, shift):
    """rowmap == split_feature + roll + merge_splits of the reference (attention.py:60-107), without the copies"""
    b, h, w, splits, c = 2, 12, 20, 2, 128
    wh, ww = h // splits, w // splits
    q, k, v = rnd(b, h * w, c), rnd(b, h * w, c), rnd(b, h * w, cv)
    idx = torch.arange(b * h * w, dtype=torch.int32).view(b, h, w)
    if shift:
        idx = torch.roll(idx, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2))
    rowmap = idx.view(b, splits, wh, splits, ww).permute(0, 1, 3, 2, 4).reshape(b * splits * splits, wh * ww).contiguous()
    region = torch.randint(0, 3, rowmap.shape, generator=G, dtype=torch.int32) if shift else None

    def win(t):                                    # the reference's formulation on float64
        t = t.double().view(b, h, w, -1)
        if shift:
            t = torch.roll(t, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2))
        return t.view(b, splits, wh, splits, ww, -1).permute(0, 1, 3, 2, 4, 5).reshape(b * splits * splits, wh * ww, -1)
    scores = torch.matmul(win(q), win(k).transpose(1, 2)) / c ** 0.5
    if shift:
        scores = scores + torch.where(region[:, :, None] != region[:, None, :], -100.0, 0.0)
    ref = torch.matmul(torch.softmax(scores, dim=-1), win(v))
    ref = ref.view(b, splits, splits, wh, ww, cv).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, cv)
    if shift:
        ref = torch.roll(ref, shifts=(wh // 2, ww // 2), dims=(1, 2))
    for nsplit in (1, 2):
        out = hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), region.cuda() if shift else None, rowmap=rowmap.cuda(), nsplit=nsplit)
        close(out, ref.reshape(b, h * w, cv), "window attention through rowmap, nsplit=%d" % nsplit, atol=5e-5, rtol=1e-4)


def test_local_corr_kernels(hip):
    b, h, w = 2, 13, 21
    f0, f1 = rnd(b, 128, h, w), rnd(b, 128, h, w)
    t0, t1 = f0.flatten(2).transpose(1, 2).contiguous(), f1.flatten(2).transpose(1, 2).contiguous()
    close(hip.local_corr_softmax(t0.cuda(), t1.cuda(), h, w, 4), og.local_correlation_softmax(f0.double(), f1.double(), 4),
          "local_correlation_softmax", atol=2e-4, rtol=1e-4)
    flow = rnd(b, 2, h, w) * 3
    close(hip.local_corr_flow(t0.cuda(), t1.cuda(), flow.cuda(), 4), og.local_correlation_with_flow(f0.double(), f1.double(), flow.double(), 4),
          "local_correlation_with_flow", atol=2e-4, rtol=1e-4)


def test_local_attn_prop(hip):
    b, h, w = 2, 9, 14
    f, flow = rnd(b, 128, h, w), rnd(b, 2, h, w) * 4
    sd = {"feature_flow_attn.q_proj.weight": rnd(128, 128) / 11, "feature_flow_attn.q_proj.bias": rnd(128) * 0.1,
          "feature_flow_attn.k_proj.weight": rnd(128, 128) / 11, "feature_flow_attn.k_proj.bias": rnd(128) * 0.1}
    ref = og.self_attn_propagation({k: v.double() for k, v in sd.items()}, f.double(), flow.double(), 1)
    tok = f.flatten(2).transpose(1, 2).contiguous().cuda()
    q = hip.linear_tokens(tok, sd["feature_flow_attn.q_proj.weight"].cuda(), sd["feature_flow_attn.q_proj.bias"].cuda())
    k = hip.linear_tokens(tok, sd["feature_flow_attn.k_proj.weight"].cuda(), sd["feature_flow_attn.k_proj.bias"].cuda())
    close(hip.local_attn_prop(q, k, flow.cuda(), 1), ref, "local window propagation", atol=1e-4, rtol=1e-4)


def test_resampling_kernels(hip):
    x = rnd(2, 3, 11, 17)
    close(hip.bilinear_resize(x.cuda(), (22, 34)), F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=True), "x2")
    close(hip.bilinear_resize(x.cuda(), (16, 32)), F.interpolate(x.double(), size=(16, 32), mode="bilinear", align_corners=True), "resize")
    fl = rnd(2, 2, 11, 17)
    want = F.interpolate(fl.double(), size=(7, 9), mode="bilinear", align_corners=True)
    want = torch.stack([want[:, 0] * 9 / 17, want[:, 1] * 7 / 11], 1)
    close(hip.bilinear_resize(fl.cuda(), (7, 9), 9 / 17, 7 / 11), want, "flow rescale")
    flow = rnd(2, 2, 11, 17) * 4
    close(hip.flow_warp(x.cuda(), flow.cuda()), og.flow_warp(x.double(), flow.double()), "flow_warp", atol=1e-4)
    mask = rnd(2, 9 * 16, 11, 17)
    close(hip.convex_upsample(fl.cuda(), mask.cuda(), 4), og.upsample_flow_with_mask(fl.double(), mask.double(), 4), "convex upsample")
    a, bw = rnd(2, 2, 11, 17) * 2, rnd(2, 2, 11, 17) * 2
    fo, bo = hip.fb_check(a.cuda(), bw.cuda())
    rf, rb = og.forward_backward_consistency_check(a.double(), bw.double())
    assert (fo.cpu() == rf.float()).float().mean() > 0.99 and (bo.cpu() == rb.float()).float().mean() > 0.99


@pytest.mark.parametrize("cv", [128, 2])
def test_attention_tokens_wide_dynamic_range(hip, cv):
    """the two-piece fp16 attention (csrc/attention16.hip) carries power-of-two scales per query row, per staged key tile and a
    running one for the value tiles: operands of any magnitude, tiles of very different magnitude, all-zero tiles first"""
    b, l = 2, 416
    q, k, v = rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, cv)
    q[0] *= 3e-5; k[0] *= 2e4                                   # tiny queries against large keys: scores of order one
    q[1, :100] *= 40.0; k[1, 200:232] *= 1e-3                   # rows / one whole key tile of another magnitude
    v[0, :64] = 0.0                                             # two all-zero value tiles first, then tiny, then huge values
    v[0, 64:128] *= 1e-12
    v[0, 300:] *= 1e9
    v[1] *= torch.logspace(-6, 6, l)[:, None]                   # the running value scale changes at nearly every tile
    scores = torch.matmul(q.double(), k.double().transpose(1, 2)) / 128 ** 0.5
    p = torch.softmax(scores, dim=-1)
    ref = torch.matmul(p, v.double())
    for nsplit in (1, 4):
        out = hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), None, nsplit=nsplit).double().cpu()
        assert_attention_bound(out, ref, scores, p, v, "nsplit=%d" % nsplit)


def assert_attention_bound(out, ref, scores, p, v, what):
    """float32 grade relative to what an output is made of (sum_j p_ij |v_jc|), with the score error (2^-22 of sum_c |q_c k_c|)
    amplified by the softmax in proportion to the score magnitude, plus the fp16 floor of a probability: 2^-40 of the row's largest
    one, times sum_j |v_jc| (attention16.hip; irrelevant unless values span ~10 decades along the keys)"""
    mag = torch.matmul(p, v.double().abs())
    smax = scores.abs().amax(dim=(1, 2), keepdim=True)
    bound = (5e-6 + 4e-7 * smax) * mag + 2e-12 * v.double().abs().sum(dim=1, keepdim=True) + 1e-300
    worst = ((out - ref).abs() / bound).max().item()
    assert worst < 1.0, "%s: error / bound = %.3f" % (what, worst)


def test_attention_rows64_wide_dynamic_range(hip):
    """the 64-channel parallax attention (values of 96 channels, row statistics, column sums) on operands of any magnitude"""
    n, w = 3, 200
    q, k = rnd(n, w, 64), rnd(n, w, 64)
    v = torch.zeros(n, w, 96)
    v[:, :, :67] = rnd(n, w, 67)
    q[0] *= 1e-4; k[0] *= 3e4
    k[1, 64:96] *= 1e-4
    v[1] *= torch.logspace(5, -5, w)[:, None]
    v[2, :96] = 0.0
    scale = 1.0 / 64
    sc = torch.matmul(q.double(), k.double().transpose(1, 2)) * scale
    p = torch.softmax(sc, dim=-1)
    ref = torch.matmul(p, v.double())
    lib = hip.lib()
    out = torch.empty(n, w, 96, device="cuda")
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    hip.check(lib.ct_attention_rows64_f32(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), None, n, w, scale, None))
    assert_attention_bound(out.double().cpu()[:, :, :67], ref[:, :, :67], sc, p, v[:, :, :67], "rows64")
    stats = torch.empty(n, w, 2, device="cuda")
    hip.check(lib.ct_attention_rows64_f32(qc.data_ptr(), kc.data_ptr(), None, None, stats.data_ptr(), n, w, scale, None))
    m_ref = sc.max(dim=-1).values
    l_ref = torch.exp(sc - m_ref[..., None]).sum(-1)
    st = stats.double().cpu()
    assert (st[..., 0] - m_ref).abs().max().item() < 2e-5 * max(1.0, m_ref.abs().max().item())
    assert ((st[..., 1] - l_ref).abs() / l_ref).max().item() < 2e-5
    colsum = torch.empty(n, w, device="cuda")
    hip.check(lib.ct_attention_colsum64_f32(qc.data_ptr(), kc.data_ptr(), stats.data_ptr(), colsum.data_ptr(), n, w, scale, None))
    assert (colsum.double().cpu() - p.sum(dim=1)).abs().max().item() < 2e-5 * max(1.0, p.sum(dim=1).max().item())


@pytest.mark.parametrize("t,k,n,act,two", [(4096, 256, 1024, 6, True), (5000, 256, 1024, 0, False), (4100, 256, 128, 6, True),
                                           (4097, 1024, 128, 0, False), (9000, 512, 128, 0, False), (4096, 256, 512, 0, True)])
def test_linear_ws16(hip, t, k, n, act, two):
    """ct_linear_ws16_f32 (weight slice resident in LDS, two fp16 pieces): both slicings against float64 torch, token counts that
    are not multiples of the 32-token tile, two-source rows, bias, GELU; K slices come back as partial slabs that
    ct_layernorm128_f32 sums"""
    x, w, b = rnd(t, k), rnd(n, k) / k ** 0.5, rnd(n)
    ref = F.linear(x.double(), w.double(), b.double())
    if act == 6:
        ref = F.gelu(ref)
    wc, xc = w.cuda(), x.cuda()
    if two:
        out = hip.linear_tokens(xc[:, :128].contiguous(), wc, b.cuda(), act=act, x2=xc[:, 128:].contiguous(), partials=True)
    else:
        out = hip.linear_tokens(xc, wc, b.cuda(), act=act, partials=True)
    assert hasattr(wc, "_ct_lin_ws16")                                  # this shape took the resident-weight kernel
    if k > 256:
        assert out.shape == (k // 256, t, n)
        g, be, res = rnd(128), rnd(128), rnd(t, 128)
        ln = hip.layernorm128(out, g.cuda(), be.cuda(), residual=res.cuda(), partials=k // 256)
        close(ln, res.double() + F.layer_norm(ref, (128,), g.double(), be.double()), "LN of the partial slabs", atol=2e-5, rtol=2e-5)
        out = out.sum(dim=0)
    close(out, ref, "linear_ws16")
    hip.set_linear_ws16(False)                                          # the LDS-tiled kernel computes the same layer
    try:
        old = hip.linear_tokens(xc, wc, b.cuda(), act=act, partials=True)
    finally:
        hip.set_linear_ws16(True)
    assert old.shape == (t, n)
    close(old, ref, "linear_split")
    wc.mul_(2.0)                                                        # the packed image follows in-place updates
    out2 = hip.linear_tokens(xc, wc, None, partials=True)
    out2 = out2.sum(dim=0) if out2.dim() == 3 else out2
    close(out2, 2 * F.linear(x.double(), w.double()), "repacked after an in-place update")


def test_linear_ws16_wide_dynamic_range(hip):
    """the result does not depend on the magnitude of a token (per-token running power-of-two scale), nor on where in the row its
    large channels sit (the scale drops in the middle of the contraction: the accumulators are rescaled)"""
    t, k, n = 4096, 256, 256
    x, w = rnd(t, k), rnd(n, k) / 16
    scale = torch.pow(2.0, torch.randint(-20, 21, (t, 1), generator=G).float())
    xs = x * scale
    xs[::3, 200:] *= 4096.0                                              # large channels late in the row
    xs[1::3, :16] *= 1e4                                                 # ... and early
    xs[5] = 0.0
    w2 = w * 1e-6                                                       # the per-layer weight scale
    for ww in (w, w2):
        ref = F.linear(xs.double(), ww.double())
        out = hip.linear_tokens(xs.cuda(), ww.cuda(), None).cpu().double()
        bound = F.linear(xs.double().abs(), ww.double().abs()) + 1e-300      # sum |x||w|: the natural error scale of a dot product
        assert ((out - ref).abs() / bound).max().item() < 1e-6
    xk = rnd(t, 1024) * scale
    xk[::2, 700:] *= 1e3
    wk = rnd(128, 1024) / 32
    out = hip.linear_tokens(xk.cuda(), wk.cuda(), None, partials=True)
    assert out.shape == (4, t, 128)
    ref = F.linear(xk.double(), wk.double())
    bound = F.linear(xk.double().abs(), wk.double().abs()) + 1e-300
    assert ((out.sum(0).cpu().double() - ref).abs() / bound).max().item() < 1e-6
    lib, p = hip.lib(), out.data_ptr()
    assert lib.ct_linear_ws16_f32(p, None, 256, p, 0, None, p, 64, 256, 96, 0, None, None, None, None) == -1      # n % 128
    assert lib.ct_linear_ws16_f32(p, None, 1024, p, 0, None, p, 64, 1024, 128, 6, None, None, None, None) == -1   # no activation on partial slabs
    assert lib.ct_linear_ws16_f32(p, None, 256, p, 0, None, p, 0, 256, 128, 0, None, None, None, None) == 0       # no tokens
    assert lib.ct_linear_ws16_f32(p, None, 256, p, 0, None, p, 64, 256, 128, 0, p, p, None, None) == -1   # LayerNorm epilogue: k = n = 128 only


@pytest.mark.parametrize("count,t,bias", [(3, 4096, False), (2, 5001, True), (1, 4100, False), (4, 8192, True)])
def test_linear_tokens_multi(hip, count, t, bias):
    """several 128 -> 128 projections of the same tokens in one launch (slab outputs of ct_linear_ws16_f32) == the separate layers"""
    x = rnd(2, t // 2 if t % 2 == 0 else t, 128) if t % 2 == 0 else rnd(t, 128)
    ws = [(rnd(128, 128) / 128 ** 0.5 * (10.0 ** i)).cuda() for i in range(count)]
    bs = [rnd(128).cuda() for _ in range(count)] if bias else None
    outs = hip.linear_tokens_multi(x.cuda(), ws, bs)
    assert hasattr(ws[0], "_ct_lin_ws16_multi") and len(outs) == count
    for i, o in enumerate(outs):
        ref = F.linear(x.double(), ws[i].double().cpu(), bs[i].double().cpu() if bias else None)
        assert o.shape == x.shape and o.is_contiguous()
        bound = F.linear(x.double().abs(), ws[i].double().cpu().abs()) + (bs[i].double().cpu().abs() if bias else 0) + 1e-300
        assert ((o.double().cpu() - ref).abs() / bound).max().item() < 1e-6
    small = hip.linear_tokens_multi(x.cuda()[..., :64, :].contiguous(), ws, bs)       # few tokens: the separate kernels
    assert len(small) == count and small[0].shape[-1] == 128


@pytest.mark.parametrize("t,res,bias", [(4096, True, False), (5003, False, True), (64, True, True)])
def test_linear_layernorm128(hip, t, res, bias):
    """merge projection + LayerNorm (+ skip) in one launch (the LayerNorm epilogue of ct_linear_ws16_f32) == the two kernels"""
    x, w = rnd(t, 128) * 3, (rnd(128, 128) / 128 ** 0.5).cuda()
    b = rnd(128) if bias else None
    g, be, r = rnd(128), rnd(128), rnd(t, 128) if res else None
    lin = F.linear(x.double(), w.double().cpu(), b.double() if bias else None)
    ref = F.layer_norm(lin, (128,), g.double(), be.double()) + (r.double() if res else 0)
    out = hip.linear_layernorm128(x.cuda(), w, b.cuda() if bias else None, g.cuda(), be.cuda(), residual=r.cuda() if res else None)
    close(out, ref, "linear + LayerNorm", atol=2e-5, rtol=2e-5)
    xs = x * torch.pow(2.0, torch.randint(-12, 13, (t, 1), generator=G).float())     # LayerNorm is scale invariant per token: so is the error
    lin = F.linear(xs.double(), w.double().cpu())
    ref = F.layer_norm(lin, (128,), g.double(), be.double())
    close(hip.linear_layernorm128(xs.cuda(), w, None, g.cuda(), be.cuda()), ref, "tokens of any magnitude", atol=3e-5, rtol=3e-5)


@pytest.mark.parametrize("l", [1, 7, 28, 33, 65])
def test_attention_tiny_sequences(hip, l):
    """fewer keys than one 32-key tile, one more than a tile, ...: ragged first / last tiles of the streaming kernels"""
    b = 3
    q, k, v, v2 = rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, 2)
    p = torch.softmax(torch.matmul(q.double(), k.double().transpose(1, 2)) / 128 ** 0.5, dim=-1)
    close(hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), None, nsplit=1), torch.matmul(p, v.double()), "cv 128", atol=5e-5, rtol=1e-4)
    close(hip.attention_tokens(q.cuda(), k.cuda(), v2.cuda(), None, nsplit=1), torch.matmul(p, v2.double()), "cv 2", atol=5e-5, rtol=1e-4)
    q6, k6 = rnd(b, l, 64), rnd(b, l, 64)
    v96 = torch.zeros(b, l, 96)
    v96[:, :, :67] = rnd(b, l, 67)
    p6 = torch.softmax(torch.matmul(q6.double(), k6.double().transpose(1, 2)) / 64, dim=-1)
    lib = hip.lib()
    out, stats, colsum = torch.empty(b, l, 96, device="cuda"), torch.empty(b, l, 2, device="cuda"), torch.empty(b, l, device="cuda")
    qc, kc, vc = q6.cuda(), k6.cuda(), v96.cuda()
    hip.check(lib.ct_attention_rows64_f32(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), None, b, l, 1.0 / 64, None))
    close(out[:, :, :67], torch.matmul(p6, v96.double())[:, :, :67], "rows64", atol=5e-5, rtol=1e-4)
    hip.check(lib.ct_attention_rows64_f32(qc.data_ptr(), kc.data_ptr(), None, None, stats.data_ptr(), b, l, 1.0 / 64, None))
    hip.check(lib.ct_attention_colsum64_f32(qc.data_ptr(), kc.data_ptr(), stats.data_ptr(), colsum.data_ptr(), b, l, 1.0 / 64, None))
    close(colsum, p6.sum(dim=1), "column sums", atol=5e-5, rtol=1e-4)


def test_space_to_depth_and_shared_stride2_input(hip):
    """ct_space_to_depth2_f32: channel (2 sy + sx) C + c of the result is x[:, c, sy::2, sx::2].  No cache in the binding (ADVICE r04:
    inference tensors have no version counter, raw-pointer writers do not bump it): a caller whose two stride-2 convolutions read
    one input passes the image to both (gconv2d(..., s2d=...)) -- same bits as without, also under torch.inference_mode()."""
    x = rnd(2, 5, 12, 24).cuda()
    s2d = hip.space_to_depth2(x)
    assert s2d.shape == (2, 20, 6, 12)
    for sy in range(2):
        for sx in range(2):
            assert torch.equal(s2d[:, (2 * sy + sx) * 5:(2 * sy + sx + 1) * 5], x[:, :, sy::2, sx::2])
    x.mul_(2.0)
    again = hip.space_to_depth2(x)
    assert again is not s2d and torch.equal(again[:, :5], x[:, :, ::2, ::2])
    view = rnd(2, 8, 12, 24).cuda()[:, 2:7]                   # a channel slice: batch stride != c * h * w
    assert torch.equal(hip.space_to_depth2(view)[:, 5:10], view[:, :, 0::2, 1::2])
    with pytest.raises(hip.CtHipError):
        hip.space_to_depth2(rnd(1, 4, 7, 24).cuda())
    # a down-sampling residual block's pair of convolutions (unimatch/backbone.py:9-45) on a shared image
    cin, cout, h, w = 64, 96, 32, 48
    xx = rnd(2, cin, h, w).cuda()
    w3, w1, b1 = (rnd(cout, cin, 3, 3) / 24).cuda(), (rnd(cout, cin, 1, 1) / 8).cuda(), rnd(cout).cuda()
    p3, p1 = hip.pack_gconv_weight(w3, torch.zeros(cout).cuda()), hip.pack_gconv_weight(w1, b1)
    assert hip.s2d_ok(xx)
    ref3, ref1 = hip.gconv2d(xx, p3[0], p3[1], cout, 3, 2, 1), hip.gconv2d(xx, p1[0], p1[1], cout, 1, 2, 0)
    shared = hip.space_to_depth2(xx)
    assert torch.equal(hip.gconv2d(xx, p3[0], p3[1], cout, 3, 2, 1, s2d=shared), ref3)
    assert torch.equal(hip.gconv2d(xx, p1[0], p1[1], cout, 1, 2, 0, s2d=shared), ref1)
    want = torch.nn.functional.conv2d(xx.double(), w3.double(), None, 2, 1)
    assert (ref3.double() - want).abs().max().item() < 1e-4
    with torch.inference_mode():                               # Lightning's test loop (the reference's `utils.cli test`)
        xi = xx.clone()
        assert torch.equal(hip.gconv2d(xi, p3[0], p3[1], cout, 3, 2, 1), ref3)
        assert torch.equal(hip.gconv2d(xi, p1[0], p1[1], cout, 1, 2, 0, s2d=hip.space_to_depth2(xi)), ref1)
    with pytest.raises(hip.CtHipError):
        hip.gconv2d(xx, p3[0], p3[1], cout, 3, 2, 1, s2d=shared[:, :8])


@pytest.mark.parametrize("kind", ["smooth", "constant", "jump", "random", "outside", "nan"])
def test_local_corr_flow_tile_form(hip, kind):
    """ct_local_corr_flow_f32 in its tile form (csrc/gmflow.hip: a 4 x 8 pixel tile shares the box of its windows in LDS, one float32
    MFMA GEMM per tile) against the reference's grid_sample formulation (oracle/gmflow.py, float64): smooth flows (the shared box),
    a flow discontinuity and random flows (boxes beyond the LDS budget: the per-pixel form inside the same launch), windows that
    leave the image partly or entirely, NaN / inf flows, ragged tiles at the right / bottom edges"""
    b, h, w = 2, 22, 43
    f0, f1 = rnd(b, 128, h, w), rnd(b, 128, h, w)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    if kind == "smooth":
        flow = torch.stack([0.07 * xx - 0.03 * yy + 0.4, 0.05 * yy + 0.02 * xx - 1.3], 0)[None].repeat(b, 1, 1, 1)
    elif kind == "constant":
        flow = torch.full((b, 2, h, w), 2.25)
        flow[:, 1] = -3.5
    elif kind == "jump":
        flow = torch.zeros(b, 2, h, w)
        flow[:, 0, :, 17:] = 9.5                                  # a motion boundary through the tiles of column 16..23
        flow[:, 1, 10:, :] = -6.25
    elif kind == "random":
        flow = rnd(b, 2, h, w) * 5
    elif kind == "outside":
        flow = torch.stack([0.9 * xx - 30.0, 0.0 * yy + 14.0], 0)[None].repeat(b, 1, 1, 1)     # windows leave the image on three sides
        flow[1] = 300.0                                            # every window far outside: all zeros
    else:
        flow = rnd(b, 2, h, w)
        flow[0, 0, 3, 5] = float("nan")
        flow[1, 1, 20, 40] = float("inf")
        flow[0, :, 8:12, 8:16] = float("nan")                     # a whole tile without a finite flow
    t0, t1 = f0.flatten(2).transpose(1, 2).contiguous(), f1.flatten(2).transpose(1, 2).contiguous()
    got = hip.local_corr_flow(t0.cuda(), t1.cuda(), flow.cuda(), 4).cpu().double()
    fin = torch.isfinite(flow).all(dim=1, keepdim=True)
    ref = og.local_correlation_with_flow(f0.double(), f1.double(), torch.where(fin, flow, torch.full_like(flow, 1e5)).double(), 4)
    assert got.shape == ref.shape == (b, 81, h, w)
    assert torch.isfinite(got).all()                               # a non-finite flow samples nothing: zeros (the clamp of the kernel)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-4, err_msg=kind)
    if kind == "outside":
        assert got[1].abs().max().item() == 0.0


@pytest.mark.parametrize("b,h,w,r", [(2, 22, 43, 4), (1, 4, 8, 4), (1, 3, 5, 2), (2, 17, 64, 3)])
def test_local_corr_softmax_tile_form(hip, b, h, w, r):
    """ct_local_corr_softmax_f32 (tile form: the tile's neighbourhood box through one float32-MFMA GEMM) against matching.py:42-86 in
    float64: images smaller than the window, ragged tiles, taps outside the image masked to -1e9"""
    f0, f1 = rnd(b, 128, h, w), rnd(b, 128, h, w)
    t0, t1 = f0.flatten(2).transpose(1, 2).contiguous(), f1.flatten(2).transpose(1, 2).contiguous()
    close(hip.local_corr_softmax(t0.cuda(), t1.cuda(), h, w, r), og.local_correlation_softmax(f0.double(), f1.double(), r),
          "local_correlation_softmax, tile form", atol=2e-4, rtol=1e-4)


def test_stride2_conv_in_a_graph_reads_fresh_data(hip):
    """a captured stride-2 convolution contains the kernel that makes its space-to-depth image: a replayed graph convolves what is in
    its input NOW"""
    x = rnd(2, 64, 24, 64).cuda()
    wt, b = (rnd(96, 64, 3, 3) / 24).cuda(), rnd(96).cuda()
    wp, bp = hip.pack_gconv_weight(wt, b)
    out = torch.empty((2, 96, 12, 32), device="cuda")
    hip.gconv2d(x, wp, bp, 96, (3, 3), 2, (1, 1), out=out)               # warm
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.gconv2d(x, wp, bp, 96, (3, 3), 2, (1, 1), out=out)
    x2 = rnd(2, 64, 24, 64).cuda()
    x.copy_(x2)
    graph.replay()
    torch.cuda.synchronize()
    close(out, F.conv2d(x2.cpu().double(), wt.cpu().double(), b.cpu().double(), stride=2, padding=1), "stride-2 conv replayed on new data")
