#!/usr/bin/env python3
"""Golden vectors for the iCID metric by RUNNING the reference's utils/icid.py (build container only):

    python3 -B tests/golden/make_golden_icid.py

utils/icid.py imports kornia.color.rgb_to_lab and torchvision...gaussian_blur, both absent offline.  They are stood in by
oracle.metrics.rgb_to_lab / gaussian_blur (restatements of those two third-party functions, "parity unpinned" for them);
every line of the reference's icid() itself (utils/icid.py:28-152: downsampling factor, bilinear resize, the eleven
blurred moments, the seven maps, weights, exponents, the mean) is executed as written.  Only data is written."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from oracle import metrics as om  # noqa: E402

for name in ("kornia", "kornia.color", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["kornia.color"].rgb_to_lab = om.rgb_to_lab
sys.modules["torchvision.transforms.functional"].gaussian_blur = om.gaussian_blur
spec = importlib.util.spec_from_file_location("ref_icid", "/root/reference/utils/icid.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def pair(seed, h, w, strength):
    """a textured ground truth and a colour-distorted copy, both on the 8-bit grid (what real frames are, and small to store)"""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(1, 3, h // 8 + 2, w // 8 + 2, generator=g)
    gt = torch.nn.functional.interpolate(base, size=(h, w), mode="bicubic", align_corners=True).clamp(0, 1)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    gt = (gt + 0.04 * torch.sin(xx / 3.0)[None, None] * torch.cos(yy / 2.0)[None, None]).clamp(0, 1)
    mix = torch.tensor([[0.9, 0.08, 0.0], [0.05, 0.8, 0.05], [0.0, 0.1, 1.05]])
    out = (torch.einsum("ij,bjhw->bihw", mix, gt) ** (1 + strength) + 0.1 * strength).clamp(0, 1)
    q = lambda t: (t * 255).round().to(torch.uint8)      # noqa: E731
    return q(out), q(gt)


def main():
    fix = {}
    for tag, (h, w, s) in {"a": (48, 64, 0.15), "b": (97, 130, 0.05), "c": (400, 520, 0.3), "d": (650, 700, 0.1)}.items():
        x8, y8 = pair(len(tag) + h, h, w, s)
        x, y = x8.float() / 255, y8.float() / 255
        fix[tag + "/x_u8"], fix[tag + "/y_u8"] = x8.numpy(), y8.numpy()
        fix[tag + "/icid_f32"] = np.float64(ref.icid(x, y))                       # the reference as it runs (float32 tensors)
        fix[tag + "/icid_f64"] = np.float64(ref.icid(x.double(), y.double()))
        fix[tag + "/icid_same"] = np.float64(ref.icid(x, x))
        print(tag, (h, w), "factor", max(1, round(min(h, w) / 256)), "iCID f32 %.8f f64 %.8f identical-images %.2e"
              % (fix[tag + "/icid_f32"], fix[tag + "/icid_f64"], fix[tag + "/icid_same"]))
    np.savez_compressed(os.path.join(OUT, "icid.npz"), torch=torch.__version__, **fix)
    print("wrote icid.npz")


if __name__ == "__main__":
    main()
