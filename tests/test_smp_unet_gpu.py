"""f4 (GPU): the EfficientNet-B2 / U-Net kernels (csrc/unet.hip) and smp_hip's forward against oracle/smp_unet.py evaluated
in float64 on the CPU (a restatement of smp / efficientnet_pytorch: parity unpinned), then DMSCT.forward with its default
modules.  Tolerances are float32 rounding level, measured x2 where stated."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F   # noqa: E402

from oracle import smp_unet as o   # noqa: E402
from oracle import gmflow as og   # noqa: E402

G = torch.Generator().manual_seed(11)


def rnd(*shape):
    return torch.randn(*shape, generator=G)


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


def close(a, b, msg, atol=2e-5, rtol=2e-5):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("k,s,pad,h,w", [(3, 1, (1, 1), 37, 70), (3, 2, (0, 1), 36, 130), (5, 2, (2, 2), 34, 66), (5, 1, (2, 2), 17, 30),
                                          (3, 2, (1, 1), 33, 65), (5, 2, (2, 2), 272, 480)])
def test_dwconv_and_tile_sums(hip, k, s, pad, h, w):
    n, c = 2, 7
    x, wt, b = rnd(n, c, h, w), rnd(c, 1, k, k) / k, rnd(c)
    ref = F.conv2d(F.pad(x.double(), (pad[0], pad[1], pad[0], pad[1])), wt.double(), b.double(), stride=s, groups=c)
    ref = ref * torch.sigmoid(ref)
    ho, wo = ref.shape[2:]
    out, sums = hip.dwconv(x.cuda(), wt.reshape(c, k * k).contiguous().cuda(), b.cuda(), k, s, (pad[0], pad[0]), (ho, wo), want_sums=True)
    close(out, ref, "depthwise conv + swish")
    assert sums.shape == (n, c, hip.lib().ct_dwconv_tiles(ho, wo))
    close(sums.sum(-1), ref.sum((2, 3)), "tile sums", atol=1e-3 * (ho * wo) ** 0.5, rtol=1e-5)
    lin = hip.dwconv(x.cuda(), wt.reshape(c, k * k).contiguous().cuda(), b.cuda(), k, s, (pad[0], pad[0]), (ho, wo), act=0)
    close(lin, F.conv2d(F.pad(x.double(), (pad[0], pad[1], pad[0], pad[1])), wt.double(), b.double(), stride=s, groups=c), "no activation")


def test_se_gate_scale_upcat_and_padded_conv(hip):
    n, c, nsq, h, w = 3, 96, 4, 20, 36
    x = rnd(n, c, h, w)
    wr, br, we, be = rnd(nsq, c) / c ** 0.5, rnd(nsq) * 0.1, rnd(c, nsq) / 2, rnd(c) * 0.1
    # tile sums of an identity 3x3 depthwise kernel = sums of x itself
    ident = torch.zeros(c, 9); ident[:, 4] = 1.0
    y, sums = hip.dwconv(x.cuda(), ident.cuda(), torch.zeros(c).cuda(), 3, 1, (1, 1), (h, w), act=0, want_sums=True)
    assert torch.equal(y.cpu(), x)
    gate = hip.se_gate(sums, h * w, wr.cuda(), br.cuda(), we.cuda(), be.cuda())
    mean = x.double().mean((2, 3))
    sq = mean @ wr.double().t() + br.double()
    sq = sq * torch.sigmoid(sq)
    want = torch.sigmoid(sq @ we.double().t() + be.double())
    close(gate, want, "SE gate", atol=1e-6, rtol=1e-5)
    z = x.cuda().clone()
    hip.scale_planes_(z, gate)
    close(z, x.double() * want[:, :, None, None], "gated planes", atol=1e-6, rtol=1e-5)
    odd = rnd(2, 5, 7, 9).cuda()                        # plane % 4 != 0: scalar path
    g2 = torch.rand(2, 5, generator=G).cuda()
    want2 = odd.cpu() * g2.cpu()[:, :, None, None]
    hip.scale_planes_(odd, g2)
    assert torch.equal(odd.cpu(), want2)
    a, skip = rnd(2, 5, 6, 10), rnd(2, 3, 12, 20)
    up = hip.upsample2_concat(a.cuda(), skip.cuda())
    assert torch.equal(up.cpu(), torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), skip], 1))
    assert torch.equal(hip.upsample2_concat(a.cuda()).cpu(), F.interpolate(a, scale_factor=2, mode="nearest"))
    # stride-2 3x3 stem with TF-"SAME" (0, 1) padding
    img, wt, b = torch.rand(2, 3, 64, 96, generator=G), rnd(32, 3, 3, 3) / 5, rnd(32) * 0.1
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    ref = F.conv2d(F.pad(img.double(), (0, 1, 0, 1)), wt.double(), b.double(), stride=2)
    got = hip.gconv2d_pad(img.cuda(), wp, bp, 32, 3, 2, (0, 0), (32, 48), act=hip.ACT_SWISH)
    close(got, ref * torch.sigmoid(ref), "padded stem conv + swish")


def _load(module, sd):
    module.load_state_dict({k: v.float() for k, v in sd.items()})
    return module.cuda().eval()


@pytest.mark.parametrize("n,h,w", [(1, 96, 160), (2, 64, 80)])
def test_encoder_decoder_head_vs_restatement(hip, n, h, w, conv_mode):
    import smp_hip
    esd = o.random_state(o.encoder_param_shapes(), 5, torch.float64)
    enc = _load(smp_hip.get_encoder("efficientnet-b2", depth=4, weights=None), esd)
    x = torch.rand(n, 3, h, w, generator=G)
    want = o.encoder_forward(esd, x.double())
    got = enc(x.cuda())
    assert len(got) == 5
    errs = []
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_.shape == w_.shape
        errs.append(float((g_.cpu().double() - w_).abs().max() / w_.abs().max()))
    print("\n[efficientnet-b2 encoder %dx%dx%d, %s convs] max error / max|feature| per scale: %s" % (n, h, w, conv_mode, ["%.1e" % e for e in errs]))
    assert max(errs) < 4e-6                                  # measured round 2: 1.2e-6 (both conv modes)
    channels = [2 * c + 1 for c in enc.out_channels]
    dsd = o.random_state(o.decoder_param_shapes(channels), 6, torch.float64)
    hsd = o.random_state(o.head_param_shapes(), 7, torch.float64)
    dec = _load(smp_hip.UnetDecoder(channels, (256, 128, 64, 32), n_blocks=4, use_batchnorm=False), dsd)
    head = _load(smp_hip.SegmentationHead(32, 3), hsd)
    fused = [torch.cat([f, 0.5 * f, f[:, :1]], 1) for f in want]
    want_y = o.head_forward(hsd, o.decoder_forward(dsd, *fused))
    got_y = head(dec(*[f.float().cuda() for f in fused]))
    e = float((got_y.cpu().double() - want_y).abs().max() / want_y.abs().max())
    print("[unet decoder + head] max error / max|output| %.1e" % e)
    assert e < 1e-5                                          # measured round 2: 4.1e-6


@pytest.mark.parametrize("h,w", [(200, 312), (540, 960)])      # 540 x 960 = BASELINE.json configs[3] at its full size
def test_dmsct_forward_default_modules(hip, golden_dir, h, w):
    """DMSCT() as the reference constructs it (configs/dmsct.yaml): matcher + EfficientNet-B2 encoder + U-Net decoder + head
    on the device, against the float64 composition of the oracle pieces fed the device matcher's flow and mask."""
    from methods.dmsct import DMSCT
    torch.manual_seed(0)
    model = DMSCT().cuda().eval()
    esd = o.random_state(o.encoder_param_shapes(), 15, torch.float64)
    channels = [2 * c + 1 for c in o.OUT_CHANNELS[:5]]
    dsd = o.random_state(o.decoder_param_shapes(channels), 16, torch.float64)
    hsd = o.random_state(o.head_param_shapes(), 17, torch.float64)
    hsd = {k: 0.2 * v for k, v in hsd.items()}                 # keep the residual inside [0, 1] for most pixels
    _load(model.encoder, esd); _load(model.decoder, dsd); _load(model.head, hsd)
    # 200 x 312 is not a multiple of 16 (replicate padding + crop are exercised); 540 pads to 544
    target, reference = torch.rand(1, 3, h, w, generator=G), torch.rand(1, 3, h, w, generator=G)
    out = model(target.cuda(), reference.cuda())
    assert out.shape == (1, 3, h, w) and torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    m = model.match(target.cuda(), reference.cuda())
    pad = og.dmsct_pad_size(reference.shape)
    ft = o.encoder_forward(esd, F.pad(target.double(), pad, mode="replicate"))
    fr = o.encoder_forward(esd, F.pad(reference.double(), pad, mode="replicate"))
    fused = og.dmsct_fuse_features(m["flow"].cpu().double(), m["fwd_occ"].cpu().double(), ft, fr, pad)
    want = torch.clamp(target.double() + o.head_forward(hsd, o.decoder_forward(dsd, *fused))[:, :, :h, :w], 0, 1)
    # the same glue lines in float32, as the reference evaluates them (flow_warp's sampling coordinates): with the huge, rough
    # flow fields of a random-weight matcher at full size their distance from float64 is the floor of any float32 pipeline
    fused32 = og.dmsct_fuse_features(m["flow"].cpu(), m["fwd_occ"].cpu(), [f.float() for f in ft], [f.float() for f in fr], pad)
    want32 = torch.clamp(target.double() + o.head_forward(hsd, o.decoder_forward(dsd, *[f.double() for f in fused32]))[:, :, :h, :w], 0, 1)
    floor = float((want32 - want).abs().max())
    e = float((out.cpu().double() - want).abs().max())
    inside = float(((want > 0) & (want < 1)).double().mean())
    print("\n[dmsct default modules %dx%d] max-abs output error %.2e (float32 glue of the reference vs float64: %.2e; %.0f%% of the "
          "pixels unclamped)" % (h, w, e, floor, 100 * inside))
    assert inside > 0.3 and e < max(6e-5, 2 * floor)         # measured: 2.7e-5 (200 x 312, round 2)
