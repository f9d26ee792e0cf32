"""The deterministic colour distortions of the reference's artificial test set (utils/data.py:12-22), torch CPU restatement.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference calls torchvision.transforms.functional.adjust_{brightness,contrast,saturation,hue,gamma} on uint8 CHW tensors;
torchvision is third-party (unpinned in requirements.txt) and absent offline, so its tensor backend
(torchvision/transforms/_functional_tensor.py: _blend, rgb_to_grayscale, adjust_*, _rgb2hsv, _hsv2rgb, convert_image_dtype) is
restated here from its published source -- PARITY UNPINNED: no fixture from the real library exists for these functions.
`setup_grid_distortions` itself (which factors, in which order) follows the reference line by line."""
import numpy as np
import torch


def _to_u8_from_float(x):                      # convert_image_dtype(float -> uint8)
    return x.mul(255.0 + 1.0 - 1e-3).to(torch.uint8)


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 255).to(img1.dtype)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def adjust_brightness(img, factor):
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    mean = torch.mean(rgb_to_grayscale(img).to(torch.float32), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, factor)


def adjust_saturation(img, factor):
    return _blend(img, rgb_to_grayscale(img), factor)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc, minc = torch.max(img, dim=-3).values, torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    div = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - (s * f)), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a4 = torch.stack((torch.stack((v, q, p, p, t, v), dim=-3), torch.stack((t, v, v, q, p, p), dim=-3),
                      torch.stack((p, p, t, v, v, q), dim=-3)), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, hue_factor):
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError("hue_factor (%s) is not in [-0.5, 0.5]." % hue_factor)
    x = _rgb2hsv(img.to(torch.float32) / 255.0)
    h, s, v = x.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _to_u8_from_float(_hsv2rgb(torch.stack((h, s, v), dim=-3)))


def adjust_gamma(img, gamma, gain=1):
    x = img.to(torch.float32) / 255.0
    return _to_u8_from_float((gain * x ** gamma).clamp(0, 1))


KINDS = {"identity": 0, "brightness": 1, "contrast": 2, "saturation": 3, "hue": 4, "gamma": 5}
_FUNCS = {"identity": lambda x, p: x, "brightness": adjust_brightness, "contrast": adjust_contrast,
          "saturation": adjust_saturation, "hue": adjust_hue, "gamma": adjust_gamma}


def setup_grid_distortions(max_magnitude=0.5, num=6):
    """utils/data.py:12-22 as (name, parameter) pairs: identity first, then per magnitude brightness, contrast, saturation,
    hue, gamma"""
    specs = [("identity", 0.0)]
    for magnitude in np.linspace(-max_magnitude, max_magnitude, num):
        specs += [("brightness", 1 + magnitude), ("contrast", 1 + magnitude), ("saturation", 1 + magnitude),
                  ("hue", magnitude), ("gamma", 1 + magnitude)]
    return specs


def apply(img_u8, name, param):
    return _FUNCS[name](img_u8, float(param))
