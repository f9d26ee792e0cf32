"""Building blocks of the DCMCS3DI backbone on HIP kernels (reference pasmnet/backbone.py:4-15).

`ResB` keeps the reference's attribute tree -- `body.0` conv3x3, `body.1` LeakyReLU, `body.2` conv3x3 -- because those
names are the state_dict contract (reference checkpoints load strictly); its arithmetic, and that of every plain
Conv2d of the network, runs in ct_conv2d_split_f32 / ct_conv2d_f32 (csrc/conv_split.hip, csrc/cnn.hip) on weights
re-packed into MFMA operand order once per parameter version.  There is no eager / CPU forward.
"""
import torch

import ct_hip


def packed_weights(conv):
    """MFMA-operand-order copy of a conv's parameters (+ the split-bf16 image), cached ON the module and rebuilt when the
    parameters change (in-place update -> _version, load_state_dict / .to() -> data_ptr).  A global cache keyed by id()
    would hand a new module the packed weights of a dead one whose id and storage were recycled."""
    w, b = conv.weight, conv.bias
    version = (w._version, w.data_ptr(), -1 if b is None else b._version, -1 if b is None else b.data_ptr(), str(w.device))
    cached = getattr(conv, "_ct_packed", None)
    if cached is None or cached[0] != version:
        cached = (version, ct_hip.pack_conv_weight(w, b))
        conv._ct_packed = cached
    return cached[1]


def conv_forward(conv, x, act=0, residual=None, clamp=False, out=None, x2=None, x3=None):
    """one Conv2d (3x3 / 1x1, stride 1, "same") with its epilogue fused: bias, LeakyReLU (act=1), skip tensor, clamp; x2 / x3:
    tensors whose channels follow x's (the conv reads torch.cat([x, x2, x3], 1) without the concatenation being built)"""
    wp, bias = packed_weights(conv)
    return ct_hip.conv2d(x, wp, bias, conv.out_channels, conv.kernel_size[0], act=act, residual=residual, clamp=clamp, out=out,
                         x2=x2, x3=x3)


def conv_forward_rows(conv, x, out=None, c0=0, channels=None):
    """conv_forward writing token rows [N*H, W, channels] (channels c0 .. c0 + cout) for the streaming attention: the split
    kernel's epilogue stores that layout directly; otherwise (exact mode, W % 4) the NCHW result is transposed"""
    wp, bias = packed_weights(conv)
    rows = ct_hip.conv2d_rows(x, wp, bias, conv.out_channels, conv.kernel_size[0], out=out, c0=c0, channels=channels,
                              raw=(conv.weight, conv.bias))
    if rows is None:
        n, _, h, w = x.shape
        if out is None:
            out = torch.empty((n * h, w, channels or conv.out_channels), dtype=torch.float32, device=x.device)
        rows = ct_hip.nchw_to_rows(conv_forward(conv, x), out, c0)
    return rows


def resb_forward(block, x, out=None):
    """x + conv(LeakyReLU(conv(x)))  (pasmnet/backbone.py:14-15): two launches, the skip rides in the second conv's epilogue"""
    hidden = conv_forward(block.body[0], x, act=1)
    return conv_forward(block.body[2], hidden, residual=x, out=out)


def _conv3x3(cin, cout):
    return torch.nn.Conv2d(cin, cout, kernel_size=3, padding=1)


class ResB(torch.nn.Module):
    """residual block; parameters live under `body.0` and `body.2` like in the reference"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        layers = [_conv3x3(in_channels, out_channels), torch.nn.LeakyReLU(inplace=True), _conv3x3(out_channels, out_channels)]
        self.body = torch.nn.Sequential(*layers)

    def forward(self, x):
        return resb_forward(self, x)
