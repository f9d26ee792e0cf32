"""DMSCT ("ours" of the reference, methods/dmsct.py:19-116): forward pass on HIP kernels.

DMSCT = frozen GMFlow matcher + `segmentation_models_pytorch` EfficientNet-B2 encoder / U-Net decoder /
segmentation head.  smp is an un-vendored third-party dependency that is absent offline: `smp_hip` implements those
three modules (same module tree and parameter names, HIP forward) against a restatement of their published structure
("parity unpinned", oracle/smp_unet.py).  Everything around them is pinned by reference-run fixtures:

  * the matcher call exactly as the reference makes it (dmsct.py:85-94): `unimatch.GMFlow` on `target*255`,
    `reference*255`, `derive_matcher_inference_size`, bidirectional flow + forward-backward occlusion;
  * the multi-scale fusion (dmsct.py:96-114): replicate padding to a multiple of 2**encoder_depth, per scale
    bilinear flow rescale, `flow_warp` of the reference features, nearest-resized `1 - fwd_occ`, concatenation;
  * the residual output (dmsct.py:116): `clamp(target + head(decoder(*features))[:, :, :H, :W], 0, 1)`.

Constructor arguments as in the reference (dmsct.py:20-25); `encoder=`, `decoder=`, `head=` optionally replace the
smp_hip modules by any torch modules with smp's calling convention (encoder(x) -> list of feature maps at strides
1,2,4,...; decoder(*features); head(x)).  Inference only (the reference's training step is out of scope).
"""
import numpy as np
import torch

import ct_hip
import smp_hip
from unimatch import GMFlow


class DMSCT(torch.nn.Module):
    def __init__(self, encoder_name="efficientnet-b2", encoder_depth=4, encoder_weights=None,
                 decoder_channels=(256, 128, 64, 32), encoder=None, decoder=None, head=None, matcher_weights=None):
        super().__init__()
        self.hparams = type("HParams", (), dict(encoder_name=encoder_name, encoder_depth=encoder_depth,
                                                encoder_weights=encoder_weights, decoder_channels=decoder_channels))()
        self.matcher = GMFlow(matcher_weights)
        for p in self.matcher.parameters():
            p.requires_grad = False                       # dmsct.py:30-32
        if encoder is None:                               # dmsct.py:34-38
            encoder = smp_hip.get_encoder(name=encoder_name, depth=encoder_depth, weights=encoder_weights)
        if decoder is None:                               # dmsct.py:40-51: every scale carries [f_t | warped f_r | 1 - occ]
            channels = [2 * c + 1 for c in encoder.out_channels]
            decoder = smp_hip.UnetDecoder(encoder_channels=channels, decoder_channels=decoder_channels, n_blocks=encoder_depth,
                                          use_batchnorm=False)
        if head is None:                                  # dmsct.py:53-56
            head = smp_hip.SegmentationHead(in_channels=decoder_channels[-1], out_channels=3)
        self.encoder, self.decoder, self.head = encoder, decoder, head

    @staticmethod
    def derive_matcher_inference_size(shape, max_area=500 * 900, padding_factor=32):     # dmsct.py:58-74
        inference_size = [int(np.ceil(shape[-2] / padding_factor)) * padding_factor,
                          int(np.ceil(shape[-1] / padding_factor)) * padding_factor]
        aspect_ratio = shape[-1] / shape[-2]
        max_h = np.floor(np.sqrt(max_area / aspect_ratio))
        max_w = np.floor(max_h * aspect_ratio)
        max_inference_size = [int(np.ceil(max_h / padding_factor)) * padding_factor,
                              int(np.ceil(max_w / padding_factor)) * padding_factor]
        if inference_size[0] * inference_size[1] > max_inference_size[0] * max_inference_size[1]:
            inference_size = max_inference_size
        return inference_size

    def derive_pad_size(self, shape):                                                      # dmsct.py:76-82
        f = 2 ** self.hparams.encoder_depth
        return [0, (shape[-1] % f != 0) * (f - shape[-1] % f), 0, (shape[-2] % f != 0) * (f - shape[-2] % f)]

    @torch.no_grad()
    def match(self, target, reference):
        """dmsct.py:85-94: the frozen matcher, flows in input resolution + occlusion masks."""
        size = DMSCT.derive_matcher_inference_size(reference.shape)
        return self.matcher(ct_hip.eltwise(4, target.float().contiguous(), s0=255.0),
                            ct_hip.eltwise(4, reference.float().contiguous(), s0=255.0),
                            inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True)

    @staticmethod
    def fuse_features(flow, fwd_occ, features_target, features_reference, pad_size):
        """dmsct.py:99-114 (D1): [f_t | flow_warp(f_r, rescaled flow) | nearest(1 - occ)] per encoder scale."""
        pad = torch.nn.functional.pad
        flow = pad(flow, pad_size, mode="replicate").contiguous()
        occ = pad(fwd_occ, pad_size, mode="replicate").contiguous()
        vis = ct_hip.eltwise(4, ct_hip.eltwise(0, ct_hip.eltwise(4, occ, s0=-1.0), torch.ones_like(occ)), s0=1.0)   # 1 - occ
        out = []
        for idx, (ft, fr) in enumerate(zip(features_target, features_reference)):
            s = 2.0 ** -idx
            size = (int(np.floor(flow.shape[2] * s)), int(np.floor(flow.shape[3] * s)))
            fl = flow if idx == 0 else ct_hip.bilinear_resize(flow, size, s, s)            # unimatch.py:84-90 (bilinear=True)
            v = vis if idx == 0 else torch.nn.functional.interpolate(vis, mode="nearest", scale_factor=s)   # index gather only
            out.append(torch.cat([ft, ct_hip.flow_warp(fr.float().contiguous(), fl), v], dim=1))
        return out

    def forward(self, target, reference):
        m = self.match(target, reference)
        _, _, height, width = reference.shape
        pad_size = self.derive_pad_size(reference.shape)
        pad = torch.nn.functional.pad
        if isinstance(self.encoder, smp_hip.EfficientNetEncoder):
            # both views in one batch (inference: every layer, the SE gate included, is per sample -- same numbers as the
            # reference's two calls, dmsct.py:98-99, half the launches)
            n = target.shape[0]
            both = self.encoder(torch.cat([pad(target, pad_size, mode="replicate"), pad(reference, pad_size, mode="replicate")], dim=0))
            ft, fr = [f[:n] for f in both], [f[n:] for f in both]
        else:
            ft = self.encoder(pad(target, pad_size, mode="replicate"))
            fr = self.encoder(pad(reference, pad_size, mode="replicate"))
        features = self.fuse_features(m["flow"], m["fwd_occ"], ft, fr, pad_size)
        return torch.clamp(target + self.head(self.decoder(*features))[:, :, :height, :width], min=0, max=1)

    @torch.no_grad()
    def test_step(self, batch, batch_idx=0, dataloader_idx=0):
        """dmsct.py:118-131,139-140 (`step(batch, "Test")`): the per-batch metrics the reference logs -- PSNR, SSIM, FSIM, iCID
        on the device (the two losses are training quantities)."""
        from methods import fsim, icid, psnr, ssim
        result = self(batch["target"], batch["reference"])
        gt = batch["gt"].to(result.device)
        return {"Test PSNR": psnr(result, gt), "Test SSIM": ssim(result, gt), "Test FSIM": fsim(result, gt), "Test iCID": icid(result, gt)}
