"""Shared by tests/golden/make_golden_gmflow.py and the GMFlow tests: a procedural, seed-free state_dict.

The GMFlow state (7.36 M parameters, 29 MB) is too big to commit, and in practice the matcher runs with
pretrained weights, so the default init is irrelevant.  Every tensor is instead derived from its NAME:
    w = randn(shape, generator seeded with crc32(name)) * gain / sqrt(fan_in)
(norm weights 1 + 0.1 randn, biases 0.1 randn).  The fixture stores only (name, shape) pairs."""
import zlib

import torch


def procedural_tensor(name, shape):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    x = torch.randn(shape, generator=g)
    if name.endswith(".bias"):
        return 0.1 * x
    if ".norm" in name and name.endswith(".weight"):
        return 1.0 + 0.1 * x
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    gain = 2.0 ** 0.5 if ("backbone" in name or "refine" in name) else 1.0
    if ".q_proj." in name or ".k_proj." in name:
        # soft attention: with unit-gain (or sharper) projections this random-weight network is chaotic -- the float32
        # reference and a float64 evaluation of the same graph then differ by tens of pixels in the final flow, so no
        # parity statement would mean anything.  At gain 0.5 the float32-vs-float64 spread is 1.4e-4 after the first
        # transformer, 2e-3 px after global matching and 5e-2 px in the final flow (measured, make_golden_gmflow.py).
        gain = 0.5
    if "flow_head.conv2" in name:
        gain = 0.1                      # small residual flows: keeps the 6 GRU iterations from amplifying rounding noise
    return x * gain / fan_in ** 0.5


def procedural_state(names, shapes):
    return {str(n): procedural_tensor(str(n), s) for n, s in zip(names, shapes)}


def test_pair(seed, h, w, shift=5):
    """A textured image and a shifted/perturbed copy, 0..255 like DMSCT feeds the matcher (dmsct.py:88-89)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(1, 3, h // 4 + 2, w // 4 + 2, generator=g)
    img0 = torch.nn.functional.interpolate(base, size=(h, w), mode="bicubic", align_corners=True).clamp(0, 1)
    img0 = (img0 + 0.15 * torch.rand(1, 3, h, w, generator=g)).clamp(0, 1)
    img1 = (torch.roll(img0, shifts=(2, shift), dims=(2, 3)) * 0.9 + 0.05).clamp(0, 1)
    return img0 * 255, img1 * 255
