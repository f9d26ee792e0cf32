"""TEST INFRASTRUCTURE (never imported by the product path).  PARITY UNPINNED.

CPU restatement, in plain torch functional calls on a state dict, of the three `segmentation_models_pytorch` pieces the
reference's DMSCT is built from (methods/dmsct.py:9-11,34-56):

  * `get_encoder("efficientnet-b2", depth=4, weights=None)` -- smp's `EfficientNetEncoder`, a subclass of
    `efficientnet_pytorch.EfficientNet` (0.7.1): MBConv blocks with TF-"SAME" *static* padding computed from the nominal
    image size 260, BatchNorm eps 1e-3, swish, squeeze-and-excitation sized from the block INPUT filters;
  * `UnetDecoder(encoder_channels, decoder_channels=(256,128,64,32), n_blocks=4, use_batchnorm=False)`;
  * `SegmentationHead(in_channels=32, out_channels=3)` (kernel 3, no upsampling, no activation).

Neither package is installed in this image and the reference's requirements.txt does not pin a version, so there is no
reference run, golden vector or source to check this file against: it is written from the published structure of smp 0.3.x
/ efficientnet_pytorch 0.7.1 (module and parameter names, `stage_idxs=(5, 8, 16, 23)`, `out_channels=(3, 32, 24, 48, 120,
352)`).  What the tests can and do establish is that the HIP implementation (color-transfer_amd/smp_hip) computes THIS
function.  Inference only (BatchNorm uses running statistics; drop-connect and dropout are identities in eval mode).
"""
import math

import torch
import torch.nn.functional as F

# efficientnet_pytorch.utils: efficientnet_params('efficientnet-b2') = (width 1.1, depth 1.2, resolution 260, dropout 0.3)
WIDTH, DEPTH, IMAGE_SIZE = 1.1, 1.2, 260
BN_EPS = 1e-3
# efficientnet_pytorch.utils.efficientnet(): r repeats, k kernel, s stride, e expand, i in, o out, se 0.25
BASE_BLOCKS = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
               (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
SE_RATIO = 0.25
STAGE_IDXS = (5, 8, 16, 23)                       # smp: encoders/efficientnet.py, "efficientnet-b2"
OUT_CHANNELS = (3, 32, 24, 48, 120, 352)


def round_filters(filters, width=WIDTH, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth=DEPTH):
    return int(math.ceil(depth * repeats))


def block_table():
    """One entry per MBConv block of EfficientNet-B2, in `_blocks` order:
    dict(k, s, e, cin, cout, nsq, image_size (nominal input size of the block: fixes the static padding))."""
    out, size = [], int(math.ceil(IMAGE_SIZE / 2))           # after the stride-2 stem
    for r, k, s, e, i, o in BASE_BLOCKS:
        cin, cout = round_filters(i), round_filters(o)
        for rep in range(round_repeats(r)):
            stride = s if rep == 0 else 1
            out.append(dict(k=k, s=stride, e=e, cin=cin if rep == 0 else cout, cout=cout,
                            nsq=max(1, int((cin if rep == 0 else cout) * SE_RATIO)), image_size=size))
            size = int(math.ceil(size / stride))
    return out


def same_pad(image_size, k, s):
    """Conv2dStaticSamePadding: (left/top, right/bottom) zero padding fixed at construction from the NOMINAL image size"""
    o = int(math.ceil(image_size / s))
    p = max((o - 1) * s + (k - 1) + 1 - image_size, 0)
    return p // 2, p - p // 2


def encoder_param_shapes():
    """name -> shape of every parameter / buffer of smp's efficientnet-b2 encoder (`_fc` is deleted by smp; the head conv
    and the blocks beyond depth 4 exist but are not used by a depth-4 encoder)"""
    sh = {}

    def bn(prefix, c):
        sh[prefix + ".weight"] = (c,); sh[prefix + ".bias"] = (c,)
        sh[prefix + ".running_mean"] = (c,); sh[prefix + ".running_var"] = (c,); sh[prefix + ".num_batches_tracked"] = ()

    stem = round_filters(32)
    sh["_conv_stem.weight"] = (stem, 3, 3, 3)
    bn("_bn0", stem)
    for idx, b in enumerate(block_table()):
        p = "_blocks.%d." % idx
        mid = b["cin"] * b["e"]
        if b["e"] != 1:
            sh[p + "_expand_conv.weight"] = (mid, b["cin"], 1, 1)
            bn(p + "_bn0", mid)
        sh[p + "_depthwise_conv.weight"] = (mid, 1, b["k"], b["k"])
        bn(p + "_bn1", mid)
        sh[p + "_se_reduce.weight"] = (b["nsq"], mid, 1, 1); sh[p + "_se_reduce.bias"] = (b["nsq"],)
        sh[p + "_se_expand.weight"] = (mid, b["nsq"], 1, 1); sh[p + "_se_expand.bias"] = (mid,)
        sh[p + "_project_conv.weight"] = (b["cout"], mid, 1, 1)
        bn(p + "_bn2", b["cout"])
    head = round_filters(1280)
    sh["_conv_head.weight"] = (head, round_filters(320), 1, 1)
    bn("_bn1", head)
    return sh


def decoder_param_shapes(encoder_channels, decoder_channels=(256, 128, 64, 32)):
    """smp UnetDecoder(use_batchnorm=False): blocks.N.conv{1,2}.0.{weight,bias}"""
    enc = list(encoder_channels)[1:][::-1]
    head = enc[0]
    ins = [head] + list(decoder_channels[:-1])
    skips = list(enc[1:]) + [0]
    sh = {}
    for i, (ci, cs, co) in enumerate(zip(ins, skips, decoder_channels)):
        sh["blocks.%d.conv1.0.weight" % i] = (co, ci + cs, 3, 3); sh["blocks.%d.conv1.0.bias" % i] = (co,)
        sh["blocks.%d.conv2.0.weight" % i] = (co, co, 3, 3); sh["blocks.%d.conv2.0.bias" % i] = (co,)
    return sh


def head_param_shapes(in_channels=32, out_channels=3):
    return {"0.weight": (out_channels, in_channels, 3, 3), "0.bias": (out_channels,)}


def random_state(shapes, seed, dtype=torch.float32):
    """A deterministic, well-conditioned random state dict for the given shapes (conv weights ~ N(0, 2/fan_in), BatchNorm
    statistics away from their defaults so that the folding is exercised)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(100, dtype=torch.long)
        elif name.endswith("running_var"):
            sd[name] = (0.5 + torch.rand(shape, generator=g)).to(dtype)
        elif name.endswith("running_mean"):
            sd[name] = (0.2 * torch.randn(shape, generator=g)).to(dtype)
        elif ("_bn" in name or ".bn" in name) and name.endswith(".weight"):
            sd[name] = (0.8 + 0.4 * torch.rand(shape, generator=g)).to(dtype)
        elif name.endswith(".bias"):
            sd[name] = (0.1 * torch.randn(shape, generator=g)).to(dtype)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = (torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dtype)
    return sd


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training=False, eps=BN_EPS)


def _swish(x):
    return x * torch.sigmoid(x)


def _same_conv(x, w, bias, k, s, image_size, groups=1):
    lo, hi = same_pad(image_size, k, s)
    return F.conv2d(F.pad(x, (lo, hi, lo, hi)), w, bias, stride=s, groups=groups)


def mbconv(sd, prefix, b, x):
    """efficientnet_pytorch.model.MBConvBlock.forward (eval)"""
    inp = x
    if b["e"] != 1:
        x = _swish(_bn(sd, prefix + "_bn0", F.conv2d(x, sd[prefix + "_expand_conv.weight"])))
    x = _swish(_bn(sd, prefix + "_bn1", _same_conv(x, sd[prefix + "_depthwise_conv.weight"], None, b["k"], b["s"], b["image_size"],
                                                    groups=x.shape[1])))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = _swish(F.conv2d(sq, sd[prefix + "_se_reduce.weight"], sd[prefix + "_se_reduce.bias"]))
    sq = F.conv2d(sq, sd[prefix + "_se_expand.weight"], sd[prefix + "_se_expand.bias"])
    x = torch.sigmoid(sq) * x
    x = _bn(sd, prefix + "_bn2", F.conv2d(x, sd[prefix + "_project_conv.weight"]))
    if b["s"] == 1 and b["cin"] == b["cout"]:
        x = x + inp
    return x


def encoder_forward(sd, x, depth=4):
    """smp EfficientNetEncoder.forward: [x, stem, blocks[:5], blocks[5:8], blocks[8:16], blocks[16:]][:depth + 1]"""
    feats = [x]
    y = _swish(_bn(sd, "_bn0", _same_conv(x, sd["_conv_stem.weight"], None, 3, 2, IMAGE_SIZE)))
    feats.append(y)
    table = block_table()
    bounds = (0,) + STAGE_IDXS
    for stage in range(2, depth + 1):
        for idx in range(bounds[stage - 2], bounds[stage - 1]):
            y = mbconv(sd, "_blocks.%d." % idx, table[idx], y)
        feats.append(y)
    return feats


def decoder_forward(sd, *features):
    """smp UnetDecoder.forward (center = Identity, attention = Identity, no BatchNorm)"""
    feats = list(features)[1:][::-1]
    x, skips = feats[0], feats[1:]
    n_blocks = len([k for k in sd if k.endswith("conv1.0.weight")])
    for i in range(n_blocks):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i < len(skips):
            x = torch.cat([x, skips[i]], dim=1)
        x = F.relu(F.conv2d(x, sd["blocks.%d.conv1.0.weight" % i], sd["blocks.%d.conv1.0.bias" % i], padding=1))
        x = F.relu(F.conv2d(x, sd["blocks.%d.conv2.0.weight" % i], sd["blocks.%d.conv2.0.bias" % i], padding=1))
    return x


def head_forward(sd, x):
    return F.conv2d(x, sd["0.weight"], sd["0.bias"], padding=1)
