#!/usr/bin/env python3
"""Per-operation golden vectors for the GMFlow matcher and the DMSCT glue, produced by calling the REFERENCE's own
functions on synthetic inputs (build container only):

    python3 -B tests/golden/make_golden_gmflow_ops.py

Why: the end-to-end goldens (make_golden_gmflow.py) run a random-weight network whose occlusion mask is all ones, so
they pin neither `forward_backward_consistency_check` nor the occlusion branch of the DMSCT glue.  Here every geometric /
matching primitive is called directly from /root/reference (unimatch/geometry.py:68-99, utils.py:137-155,
matching.py:10-126, attention.py:169-256) on inputs built so that masks are MIXED (10-60 % occluded), and
`DMSCT.forward` itself (methods/dmsct.py:84-116) is executed with its matcher replaced by a stand-in that returns a
synthetic flow / occlusion pair and with stub smp encoder / decoder / head modules (segmentation_models_pytorch is absent
offline): the decoder stub records the fused multi-scale features the real glue lines hand to it.  Only data is written.
"""
import inspect
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))           # tests/
sys.path.insert(0, "/root/reference")

# ---- stubs for what is absent offline (none of it takes part in the lines exercised here) ----------------------------
pl = types.ModuleType("pytorch_lightning")


class _LM(torch.nn.Module):
    def save_hyperparameters(self):
        f = inspect.currentframe().f_back
        self.hparams = types.SimpleNamespace(**{k: v for k, v in f.f_locals.items() if k not in ("self", "__class__")})

    def log(self, *a, **k):
        pass


pl.LightningModule = _LM
sys.modules["pytorch_lightning"] = pl
for name in ("piq", "kornia", "kornia.losses", "kornia.color", "kornia.metrics", "torchvision", "torchvision.transforms",
             "torchvision.transforms.functional", "torchvision.utils", "wandb", "segmentation_models_pytorch",
             "segmentation_models_pytorch.base", "segmentation_models_pytorch.decoders", "segmentation_models_pytorch.decoders.unet",
             "segmentation_models_pytorch.decoders.unet.decoder", "segmentation_models_pytorch.encoders"):
    sys.modules[name] = types.ModuleType(name)
for attr in ("psnr", "ssim", "fsim"):
    setattr(sys.modules["piq"], attr, None)
sys.modules["kornia.losses"].ssim_loss = None
sys.modules["kornia.color"].rgb_to_lab = None
sys.modules["kornia.metrics"].ssim = None
sys.modules["torchvision.transforms.functional"].gaussian_blur = None
sys.modules["torchvision.utils"].make_grid = None
sys.modules["kornia"].color = sys.modules["kornia.color"]
sys.modules["kornia"].metrics = sys.modules["kornia.metrics"]
sys.modules["kornia"].losses = sys.modules["kornia.losses"]

ENC_CHANNELS = (3, 32, 24, 48, 120)                 # smp EfficientNet-B2 out_channels at depth 4


class StubEncoder(torch.nn.Module):
    """deterministic feature pyramid: scale i = avg_pool(x, 2^i) mixed into ENC_CHANNELS[i] channels by a fixed matrix"""
    out_channels = ENC_CHANNELS

    def forward(self, x):
        feats = [x]
        for i in range(1, len(ENC_CHANNELS)):
            g = torch.Generator().manual_seed(100 + i)
            mix = torch.randn(ENC_CHANNELS[i], 3, generator=g)
            feats.append(torch.sin(torch.einsum("oc,bchw->bohw", mix, F.avg_pool2d(x, 2 ** i)) * 3.0))
        return feats


class StubDecoder(torch.nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.kw = kw
        self.seen = None

    def forward(self, *features):
        self.seen = [f.detach().clone() for f in features]
        return features[0][:, :3]


class StubHead(torch.nn.Module):
    def __init__(self, **kw):
        super().__init__()

    def forward(self, x):
        return 0.25 * x - 0.05


sys.modules["segmentation_models_pytorch.base"].SegmentationHead = StubHead
sys.modules["segmentation_models_pytorch.decoders.unet.decoder"].UnetDecoder = StubDecoder
sys.modules["segmentation_models_pytorch.encoders"].get_encoder = lambda name, depth, weights: StubEncoder()

torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": {}}
from unimatch.geometry import flow_warp, forward_backward_consistency_check  # noqa: E402
from unimatch.utils import upsample_flow_with_mask  # noqa: E402
from unimatch.matching import global_correlation_softmax, local_correlation_softmax, local_correlation_with_flow  # noqa: E402
from unimatch.attention import SelfAttnPropagation  # noqa: E402
from methods.dmsct import DMSCT  # noqa: E402
from gmflow_common import procedural_tensor  # noqa: E402


def smooth_flow(gen, b, h, w, mag):
    """low-frequency random flow field of roughly `mag` pixels"""
    coarse = torch.randn(b, 2, max(h // 8, 2), max(w // 8, 2), generator=gen)
    return F.interpolate(coarse, size=(h, w), mode="bicubic", align_corners=True) * mag


def consistent_pair(gen, b, h, w, mag, bad_frac):
    """(fwd, bwd) flows: bwd = -fwd sampled at the forward target (consistent), then a block of the image is given an
    unrelated backward flow -> that part fails the forward-backward check"""
    fwd = smooth_flow(gen, b, h, w, mag)
    bwd = -fwd
    for _ in range(8):                               # fixed point of bwd(y) = -fwd(y + bwd(y)): the exact inverse flow
        bwd = -flow_warp(fwd, bwd)
    bad = torch.zeros(b, 1, h, w, dtype=torch.bool)
    bad[:, :, : int(h * bad_frac * 1.6), : int(w * 0.62)] = True
    noise = torch.randn(b, 2, h, w, generator=gen) * 6
    return fwd, torch.where(bad, bwd + noise, bwd)


def main():
    fix = {}
    gen = torch.Generator().manual_seed(11)

    # ---- geometry.py:68-75 flow_warp ---------------------------------------------------------------------------------
    feat = torch.randn(2, 5, 11, 17, generator=gen)
    flow = torch.randn(2, 2, 11, 17, generator=gen) * 4
    fix["warp/feature"], fix["warp/flow"], fix["warp/out"] = feat.numpy(), flow.numpy(), flow_warp(feat, flow).numpy()

    # ---- geometry.py:78-99 forward_backward_consistency_check, mixed masks ------------------------------------------------
    for tag, (h, w, mag, bad) in {"fb_a": (24, 40, 2.0, 0.25), "fb_b": (33, 52, 2.5, 0.18)}.items():
        fwd, bwd = consistent_pair(gen, 2, h, w, mag, bad)
        fo, bo = forward_backward_consistency_check(fwd, bwd)
        fix[tag + "/fwd"], fix[tag + "/bwd"], fix[tag + "/fwd_occ"], fix[tag + "/bwd_occ"] = fwd.numpy(), bwd.numpy(), fo.numpy(), bo.numpy()
        # distance of every pixel's test statistic from its threshold (so that a comparison can skip knife-edge pixels)
        mag_ = torch.norm(fwd, dim=1) + torch.norm(bwd, dim=1)
        fix[tag + "/fwd_margin"] = (torch.norm(fwd + flow_warp(bwd, fwd), dim=1) - (0.01 * mag_ + 0.5)).numpy()
        fix[tag + "/bwd_margin"] = (torch.norm(bwd + flow_warp(fwd, bwd), dim=1) - (0.01 * mag_ + 0.5)).numpy()
        print(tag, "occluded fraction fwd %.3f bwd %.3f" % (fo.mean(), bo.mean()))
        assert 0.1 < fo.mean() < 0.6 and 0.1 < bo.mean() < 0.6

    # ---- utils.py:137-155 convex upsampling --------------------------------------------------------------------------
    fl = torch.randn(2, 2, 9, 13, generator=gen) * 2
    mask = torch.randn(2, 9 * 16, 9, 13, generator=gen)
    fix["up/flow"], fix["up/mask"], fix["up/out"] = fl.numpy(), mask.numpy(), upsample_flow_with_mask(fl, mask, 4).numpy()

    # ---- matching.py:10-126 ---------------------------------------------------------------------------------------------
    b, c, h, w = 1, 128, 9, 14
    f0, f1 = torch.randn(b, c, h, w, generator=gen), torch.randn(b, c, h, w, generator=gen)
    fix["corr/f0"], fix["corr/f1"] = f0.numpy(), f1.numpy()
    gflow, gprob = global_correlation_softmax(f0, f1, pred_bidir_flow=True)
    fix["corr/global_flow"] = gflow.numpy()
    lflow, lprob = local_correlation_softmax(f0, f1, 4)
    fix["corr/local_flow"] = lflow.numpy()
    cflow = torch.randn(b, 2, h, w, generator=gen) * 3
    fix["corr/flow_in"] = cflow.numpy()
    fix["corr/with_flow"] = local_correlation_with_flow(f0, f1, cflow, 4).numpy()

    # ---- attention.py:169-256 SelfAttnPropagation, global and 3x3 window ------------------------------------------------
    prop = SelfAttnPropagation(in_channels=128)
    sd = {k: procedural_tensor("feature_flow_attn." + k, v.shape) for k, v in prop.state_dict().items()}
    prop.load_state_dict(sd, strict=True)
    pflow = torch.randn(b, 2, h, w, generator=gen) * 4
    fix["prop/feature"], fix["prop/flow"] = f0.numpy(), pflow.numpy()
    with torch.no_grad():
        fix["prop/global"] = prop(f0, pflow).numpy()
        fix["prop/local_r1"] = prop(f0, pflow, local_window_attn=True, local_window_radius=1).numpy()

    # ---- methods/dmsct.py:84-116: the real DMSCT.forward around a stand-in matcher -----------------------------------------
    model = DMSCT()
    real_matcher = model.matcher

    for tag, (bb, hh, ww) in {"glue_a": (1, 70, 100), "glue_b": (2, 64, 96)}.items():
        target, reference = torch.rand(bb, 3, hh, ww, generator=gen), torch.rand(bb, 3, hh, ww, generator=gen)
        mflow = smooth_flow(gen, bb, hh, ww, 5.0)
        mocc = (torch.rand(bb, 1, hh // 6 + 1, ww // 6 + 1, generator=gen) > 0.65).float()
        mocc = F.interpolate(mocc, size=(hh, ww), mode="nearest")
        seen_call = {}

        class StandIn(torch.nn.Module):
            upsample_flow = real_matcher.upsample_flow      # the reference's own method (unimatch.py:84-96)

            def forward(self, img0, img1, **kw):
                seen_call.update(kw, img0_max=float(img0.max()))
                return {"flow": mflow.clone(), "fwd_occ": mocc.clone()}

        model.matcher = StandIn()
        with torch.no_grad():
            out = model(target, reference)
        assert seen_call["pred_bidir_flow"] and seen_call["fwd_bwd_consistency_check"] and seen_call["img0_max"] > 2
        fix[tag + "/target"], fix[tag + "/reference"] = target.numpy(), reference.numpy()
        fix[tag + "/flow"], fix[tag + "/fwd_occ"] = mflow.numpy(), mocc.numpy()
        fix[tag + "/inference_size"] = np.array(seen_call["inference_size"])
        fix[tag + "/pad_size"] = np.array(model.derive_pad_size(reference.shape))
        for i, f in enumerate(model.decoder.seen):
            fix[tag + "/fused_%d" % i] = f.numpy()
        fix[tag + "/out"] = out.numpy()
        print(tag, "pad", model.derive_pad_size(reference.shape), "inference size", seen_call["inference_size"],
              "occ frac %.3f" % mocc.mean(), "fused shapes", [tuple(f.shape) for f in model.decoder.seen])
    fix["enc_channels"] = np.array(ENC_CHANNELS)
    for (hh, ww) in ((540, 960), (1080, 1920), (135, 240), (96, 128), (2160, 3840), (480, 270)):
        fix["size/%dx%d" % (hh, ww)] = np.array(DMSCT.derive_matcher_inference_size((1, 3, hh, ww)))

    np.savez_compressed(os.path.join(OUT, "gmflow_ops.npz"), torch=torch.__version__, **fix)
    print("wrote gmflow_ops.npz:", sum(v.size for v in fix.values()), "values")


if __name__ == "__main__":
    main()
