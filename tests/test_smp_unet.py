"""f4 (CPU): the restated EfficientNet-B2 / U-Net structure (oracle/smp_unet.py, parity unpinned) and the host logic of
smp_hip -- module tree and parameter names, static TF-"SAME" padding, BatchNorm folding."""
import math

import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F   # noqa: E402

from oracle import smp_unet as o   # noqa: E402


def test_efficientnet_b2_tables():
    t = o.block_table()
    assert len(t) == 23 and o.STAGE_IDXS[-1] == 23
    # published EfficientNet-B2 figures: 9.11 M parameters with the 1408 -> 1000 classifier, i.e. 7 700 994 without it
    sh = o.encoder_param_shapes()
    n = sum(math.prod(s) for k, s in sh.items() if "running" not in k and "num_batches" not in k)
    assert n == 7_700_994
    assert [b["cout"] for b in t][4] == 24 and t[7]["cout"] == 48 and t[15]["cout"] == 120 and t[22]["cout"] == 352   # = smp out_channels
    # static padding from the nominal 260-pixel input: the stride-2 3x3 convolutions on even nominal sizes pad (0, 1)
    assert o.same_pad(o.IMAGE_SIZE, 3, 2) == (0, 1)
    assert o.same_pad(t[2]["image_size"], 3, 2) == (0, 1) and o.same_pad(t[5]["image_size"], 5, 2) == (2, 2)
    assert o.same_pad(t[8]["image_size"], 3, 2) == (1, 1)


def test_module_tree_matches_the_restatement():
    import smp_hip
    enc = smp_hip.get_encoder("efficientnet-b2", depth=4, weights=None)
    assert tuple(enc.out_channels) == o.OUT_CHANNELS[:5]
    want = o.encoder_param_shapes()
    got = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    channels = [2 * c + 1 for c in enc.out_channels]          # methods/dmsct.py:41-44
    dec = smp_hip.UnetDecoder(encoder_channels=channels, decoder_channels=(256, 128, 64, 32), n_blocks=4, use_batchnorm=False)
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == o.decoder_param_shapes(channels)
    head = smp_hip.SegmentationHead(in_channels=32, out_channels=3)
    assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == o.head_param_shapes()
    for blk, row in zip(enc._blocks, o.block_table()):
        assert blk.pad == o.same_pad(row["image_size"], row["k"], row["s"])
        assert (blk.cin, blk.cout, blk.k, blk.s, blk.e) == (row["cin"], row["cout"], row["k"], row["s"], row["e"])
    # the reference's own error behaviour for arguments it never uses
    with pytest.raises(ValueError):
        smp_hip.UnetDecoder(channels, (256, 128), n_blocks=4, use_batchnorm=False)
    with pytest.raises(KeyError):
        smp_hip.get_encoder("resnet-9000")
    # a state dict with the classifier smp deletes still loads
    sd = dict(enc.state_dict())
    sd["_fc.weight"], sd["_fc.bias"] = torch.zeros(1000, 1408), torch.zeros(1000)
    enc.load_state_dict(sd)


def test_batchnorm_folding():
    import smp_hip
    g = torch.Generator().manual_seed(3)
    bn = torch.nn.BatchNorm2d(6, eps=1e-3).double().eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(6, generator=g) + 0.5); bn.bias.copy_(torch.randn(6, generator=g))
        bn.running_mean.copy_(torch.randn(6, generator=g)); bn.running_var.copy_(torch.rand(6, generator=g) + 0.3)
    w = torch.randn(6, 4, 3, 3, generator=g, dtype=torch.float64)
    x = torch.randn(2, 4, 9, 11, generator=g, dtype=torch.float64)
    wf, bf = smp_hip._fold(w, bn)
    want = bn(F.conv2d(x, w, padding=1))
    got = F.conv2d(x, wf.double(), bf.double(), padding=1)
    assert (got - want).abs().max() < 1e-6          # the folded operands are float32


def test_oracle_shapes_and_dmsct_default_modules():
    sd = o.random_state(o.encoder_param_shapes(), 0, torch.float64)
    x = torch.rand(1, 3, 64, 96, dtype=torch.float64)
    feats = o.encoder_forward(sd, x)
    assert [tuple(f.shape) for f in feats] == [(1, 3, 64, 96), (1, 32, 32, 48), (1, 24, 16, 24), (1, 48, 8, 12), (1, 120, 4, 6)]
    fused = [torch.cat([f, f, f[:, :1]], 1) for f in feats]
    channels = [2 * c + 1 for c in o.OUT_CHANNELS[:5]]
    dsd = o.random_state(o.decoder_param_shapes(channels), 1, torch.float64)
    y = o.head_forward(o.random_state(o.head_param_shapes(), 2, torch.float64), o.decoder_forward(dsd, *fused))
    assert tuple(y.shape) == (1, 3, 64, 96)
    from methods.dmsct import DMSCT
    m = DMSCT()                                               # the reference's defaults (configs/dmsct.yaml)
    keys = list(m.state_dict())
    assert "encoder._blocks.22._bn2.running_var" in keys and "decoder.blocks.3.conv2.0.bias" in keys and "head.0.weight" in keys
    assert all(not p.requires_grad for p in m.matcher.parameters())
