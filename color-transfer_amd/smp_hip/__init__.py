"""The three `segmentation_models_pytorch` pieces DMSCT is built from (reference methods/dmsct.py:9-11,34-56), on HIP kernels:

    get_encoder("efficientnet-b2", depth=4, weights=None)      smp.encoders.get_encoder
    UnetDecoder(encoder_channels, decoder_channels, n_blocks, use_batchnorm=False)    smp.decoders.unet.decoder.UnetDecoder
    SegmentationHead(in_channels, out_channels)                 smp.base.SegmentationHead

Same constructor arguments, calling convention (`encoder(x)` -> list of feature maps, `decoder(*features)`, `head(x)`),
module tree and parameter names as smp 0.3.x / efficientnet_pytorch 0.7.1, so a reference checkpoint's `encoder.*`,
`decoder.*`, `head.*` entries load with `load_state_dict`; the parameters live in ordinary nn.Conv2d / nn.BatchNorm2d
holders (default initialisation like the reference's `encoder_weights=None`).  `forward` is inference only: BatchNorm
(running statistics, eps 1e-3) is folded into the convolution in front of it, drop-connect / dropout are identities.

smp and efficientnet_pytorch are third-party and not installed offline; oracle/smp_unet.py restates the same published
structure for the parity tests ("parity unpinned").  No CPU fallback: tensors must be float32 CUDA.
"""
import math

import torch
from torch import nn

import ct_hip
from ct_hip import ACT_NONE, ACT_RELU, ACT_SWISH

__all__ = ["get_encoder", "UnetDecoder", "SegmentationHead", "EfficientNetEncoder"]

# efficientnet_pytorch.utils: efficientnet_params(name) = (width, depth, resolution, dropout); smp encoders/efficientnet.py:
# out_channels, stage_idxs
_EFFICIENTNETS = {
    "efficientnet-b0": dict(width=1.0, depth=1.0, image_size=224, out_channels=(3, 32, 24, 40, 112, 320), stage_idxs=(3, 5, 9, 16)),
    "efficientnet-b1": dict(width=1.0, depth=1.1, image_size=240, out_channels=(3, 32, 24, 40, 112, 320), stage_idxs=(5, 8, 16, 23)),
    "efficientnet-b2": dict(width=1.1, depth=1.2, image_size=260, out_channels=(3, 32, 24, 48, 120, 352), stage_idxs=(5, 8, 16, 23)),
}
_BASE_BLOCKS = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
                (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]      # repeats, kernel, stride, expand, in, out (se_ratio 0.25)
_BN_EPS, _BN_MOMENTUM = 1e-3, 0.01


def _round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def _same_pad(image_size, k, s):
    """Conv2dStaticSamePadding: (top/left, bottom/right) zeros, fixed from the NOMINAL image size at construction"""
    o = int(math.ceil(image_size / s))
    p = max((o - 1) * s + (k - 1) + 1 - image_size, 0)
    return p // 2, p - p // 2


def _versions(*tensors):
    return tuple((t._version, t.data_ptr()) for t in tensors if t is not None)


def _fold(conv_w, bn):
    """conv (no bias) followed by an eval-mode BatchNorm -> (weight, bias) of the equivalent convolution"""
    g = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    w = conv_w.detach().double() * g.view(-1, 1, 1, 1)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * g
    return w.float(), b.float()


def _check_input(x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise ct_hip.CtHipError("smp_hip modules take float32 CUDA tensors [N, C, H, W] (there is no CPU path)")
    return x.contiguous()


class MBConvBlock(nn.Module):
    """efficientnet_pytorch.model.MBConvBlock (parameter holders + the HIP forward)"""

    def __init__(self, cin, cout, k, s, e, image_size):
        super().__init__()
        self.cin, self.cout, self.k, self.s, self.e = cin, cout, k, s, e
        mid = cin * e
        self.mid = mid
        if e != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOMENTUM)
        self._depthwise_conv = nn.Conv2d(mid, mid, k, stride=s, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOMENTUM)
        nsq = max(1, int(cin * 0.25))
        self._se_reduce = nn.Conv2d(mid, nsq, 1)
        self._se_expand = nn.Conv2d(nsq, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, eps=_BN_EPS, momentum=_BN_MOMENTUM)
        self.pad = _same_pad(image_size, k, s)
        self._packed = None

    def _operands(self):
        ver = _versions(*[p for p in self.parameters()], *[b for b in self.buffers()])
        if self._packed is None or self._packed[0] != ver:
            ops = {}
            if self.e != 1:
                ops["expand"] = ct_hip.pack_gconv_weight(*_fold(self._expand_conv.weight, self._bn0))
            w, b = _fold(self._depthwise_conv.weight, self._bn1)
            ops["dw"] = (w.reshape(self.mid, self.k * self.k).contiguous(), b.contiguous())
            ops["se"] = (self._se_reduce.weight.detach().float().reshape(-1, self.mid).contiguous(), self._se_reduce.bias.detach().float().contiguous(),
                         self._se_expand.weight.detach().float().reshape(self.mid, -1).contiguous(), self._se_expand.bias.detach().float().contiguous())
            ops["project"] = ct_hip.pack_gconv_weight(*_fold(self._project_conv.weight, self._bn2))
            self._packed = (ver, ops)
        return self._packed[1]

    @torch.no_grad()
    def forward(self, inputs, drop_connect_rate=None):      # model.py MBConvBlock.forward, eval mode
        ops = self._operands()
        x = inputs
        if self.e != 1:
            x = ct_hip.gconv2d(x, *ops["expand"], self.mid, 1, act=ACT_SWISH)
        h, w = x.shape[2:]
        lo, hi = self.pad
        out_size = ((h + lo + hi - self.k) // self.s + 1, (w + lo + hi - self.k) // self.s + 1)
        x, sums = ct_hip.dwconv(x, *ops["dw"], self.k, self.s, (lo, lo), out_size, act=ACT_SWISH, want_sums=True)
        gate = ct_hip.se_gate(sums, out_size[0] * out_size[1], *ops["se"])
        ct_hip.scale_planes_(x, gate)
        skip = self.s == 1 and self.cin == self.cout
        return ct_hip.gconv2d(x, *ops["project"], self.cout, 1, act=ACT_NONE, residual=inputs if skip else None)


class EfficientNetEncoder(nn.Module):
    """smp.encoders.efficientnet.EfficientNetEncoder (a subclass of efficientnet_pytorch.EfficientNet without `_fc`)"""

    def __init__(self, name="efficientnet-b2", depth=5, in_channels=3):
        super().__init__()
        if name not in _EFFICIENTNETS:
            raise KeyError("Wrong encoder name `%s`, supported encoders: %s" % (name, list(_EFFICIENTNETS)))
        if in_channels != 3:
            raise NotImplementedError("only RGB inputs (in_channels=3) are implemented")
        cfg = _EFFICIENTNETS[name]
        self._depth, self._stage_idxs, self._out_channels = depth, cfg["stage_idxs"], cfg["out_channels"]
        width, dmul, size = cfg["width"], cfg["depth"], cfg["image_size"]
        stem = _round_filters(32, width)
        self._conv_stem = nn.Conv2d(3, stem, 3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(stem, eps=_BN_EPS, momentum=_BN_MOMENTUM)
        self._stem_pad = _same_pad(size, 3, 2)
        size = int(math.ceil(size / 2))
        blocks = []
        for r, k, s, e, i, o in _BASE_BLOCKS:
            cin, cout = _round_filters(i, width), _round_filters(o, width)
            for rep in range(int(math.ceil(dmul * r))):
                stride = s if rep == 0 else 1
                blocks.append(MBConvBlock(cin if rep == 0 else cout, cout, k, stride, e, size))
                size = int(math.ceil(size / stride))
        self._blocks = nn.ModuleList(blocks)
        head = _round_filters(1280, width)                   # present in the reference's state dict, unused by the encoder
        self._conv_head = nn.Conv2d(_round_filters(320, width), head, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(head, eps=_BN_EPS, momentum=_BN_MOMENTUM)
        self._packed = None

    @property
    def out_channels(self):
        return self._out_channels[: self._depth + 1]

    @torch.no_grad()
    def forward(self, x):
        x = _check_input(x)
        ver = _versions(self._conv_stem.weight, *self._bn0.parameters(), *self._bn0.buffers())
        if self._packed is None or self._packed[0] != ver:
            self._packed = (ver, ct_hip.pack_gconv_weight(*_fold(self._conv_stem.weight, self._bn0)))
        features = [x]
        if self._depth >= 1:
            lo, hi = self._stem_pad
            h, w = x.shape[2:]
            out_size = ((h + lo + hi - 3) // 2 + 1, (w + lo + hi - 3) // 2 + 1)
            y = ct_hip.gconv2d_pad(x, *self._packed[1], self._conv_stem.out_channels, 3, 2, (lo, lo), out_size, act=ACT_SWISH)
            features.append(y)
        bounds = (0,) + tuple(self._stage_idxs[:-1]) + (len(self._blocks),)
        for stage in range(2, self._depth + 1):
            for idx in range(bounds[stage - 2], bounds[stage - 1]):
                y = self._blocks[idx](y)
            features.append(y)
        return features

    def load_state_dict(self, state_dict, **kwargs):        # smp pops the classifier it deleted
        state_dict = dict(state_dict)
        state_dict.pop("_fc.bias", None)
        state_dict.pop("_fc.weight", None)
        return super().load_state_dict(state_dict, **kwargs)


def get_encoder(name, in_channels=3, depth=5, weights=None, output_stride=32, **kwargs):
    """smp.encoders.get_encoder for the EfficientNet family the reference uses (methods/dmsct.py:34-38)"""
    if weights is not None:
        raise ValueError("pretrained encoder weights cannot be downloaded offline; load a state dict instead (weights=None)")
    if output_stride != 32:
        raise NotImplementedError("dilated encoders are not implemented")
    return EfficientNetEncoder(name, depth=depth, in_channels=in_channels)


class _Conv2dReLU(nn.Sequential):
    """smp.base.modules.Conv2dReLU with use_batchnorm=False: (conv with bias, Identity, ReLU)"""

    def __init__(self, cin, cout):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=1, bias=True), nn.Identity(), nn.ReLU(inplace=True))
        self._packed = None

    @torch.no_grad()
    def forward(self, x):
        conv = self[0]
        ver = _versions(conv.weight, conv.bias)
        if self._packed is None or self._packed[0] != ver:
            self._packed = (ver, ct_hip.pack_gconv_weight(conv.weight, conv.bias))
        return ct_hip.gconv2d(x, *self._packed[1], conv.out_channels, 3, padding=1, act=ACT_RELU)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels):
        super().__init__()
        self.conv1 = _Conv2dReLU(in_channels + skip_channels, out_channels)
        self.attention1 = nn.Identity()
        self.conv2 = _Conv2dReLU(out_channels, out_channels)
        self.attention2 = nn.Identity()

    def forward(self, x, skip=None):                          # decoder.py DecoderBlock.forward
        x = ct_hip.upsample2_concat(x, skip)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    """smp.decoders.unet.decoder.UnetDecoder (use_batchnorm=False, no attention, no center block: what DMSCT uses)"""

    def __init__(self, encoder_channels, decoder_channels, n_blocks=5, use_batchnorm=True, attention_type=None, center=False):
        super().__init__()
        if n_blocks != len(decoder_channels):
            raise ValueError("Model depth is {}, but you provide `decoder_channels` for {} blocks.".format(n_blocks, len(decoder_channels)))
        if use_batchnorm or attention_type is not None or center:
            raise NotImplementedError("only use_batchnorm=False, attention_type=None, center=False (DMSCT's decoder) is implemented")
        enc = list(encoder_channels)[1:][::-1]
        ins = [enc[0]] + list(decoder_channels[:-1])
        skips = list(enc[1:]) + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList([DecoderBlock(i, s, o) for i, s, o in zip(ins, skips, decoder_channels)])

    def forward(self, *features):
        feats = [_check_input(f) for f in features[1:]][::-1]
        x, skips = feats[0], feats[1:]
        for i, block in enumerate(self.blocks):
            x = block(x, skips[i] if i < len(skips) else None)
        return x


class SegmentationHead(nn.Sequential):
    """smp.base.SegmentationHead: (Conv2d, upsampling = Identity, activation = Identity)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, activation=None, upsampling=1):
        if activation is not None or upsampling != 1 or kernel_size != 3:
            raise NotImplementedError("only kernel_size=3, activation=None, upsampling=1 (DMSCT's head) is implemented")
        super().__init__(nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.Identity(), nn.Identity())
        self._packed = None

    @torch.no_grad()
    def forward(self, x):
        conv = self[0]
        ver = _versions(conv.weight, conv.bias)
        if self._packed is None or self._packed[0] != ver:
            self._packed = (ver, ct_hip.pack_gconv_weight(conv.weight, conv.bias))
        return ct_hip.gconv2d(_check_input(x), *self._packed[1], conv.out_channels, 3, padding=1, act=ACT_NONE)
