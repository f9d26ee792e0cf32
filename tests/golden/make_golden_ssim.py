#!/opt/conda/bin/python3.9 -B
"""Anchor for the SSIM restatement (oracle/metrics.py: piq.ssim is absent offline): scikit-image 0.18.3's
structural_similarity with Gaussian weights is the same published algorithm (Wang et al. 2004: 11x11 sigma-1.5 window,
K1 0.01, K2 0.03, population covariances, mean over the positions whose window fits) as piq's default -- run here on small
float64 images, no downsampling (min(H, W) < 384):

    /opt/conda/bin/python3.9 -B tests/golden/make_golden_ssim.py

Only data is written."""
import os

import numpy as np
from skimage.metrics import structural_similarity
import skimage

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(5)
    fix = {"skimage": skimage.__version__}
    for tag, (h, w, noise) in {"a": (40, 56, 0.05), "b": (64, 33, 0.2), "c": (120, 150, 0.02)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        gt = np.stack([0.5 + 0.4 * np.sin(xx / 7.0) * np.cos(yy / 5.0), xx / w * 0.9, 0.5 + 0.45 * np.cos((xx + yy) / 9.0)], 0)
        gt = np.clip(gt + 0.03 * rng.standard_normal(gt.shape), 0, 1)
        x = np.clip(gt ** 1.2 + noise * rng.standard_normal(gt.shape), 0, 1)
        x, gt = x.astype(np.float32).astype(np.float64), gt.astype(np.float32).astype(np.float64)   # stored as float32
        vals = [structural_similarity(x[c], gt[c], gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=1.0)
                for c in range(3)]
        fix[tag + "/x"], fix[tag + "/y"], fix[tag + "/ssim"] = x.astype(np.float32), gt.astype(np.float32), np.float64(np.mean(vals))
        print(tag, (h, w), "SSIM %.8f" % np.mean(vals))
    np.savez_compressed(os.path.join(OUT, "ssim_anchor.npz"), **fix)


if __name__ == "__main__":
    main()
