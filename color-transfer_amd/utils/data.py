"""Synthetic stand-in for the reference's DataModule (utils/data.py:128-179): the Kaggle / MSU datasets
are not available offline, and file I/O is outside the hot path.  Frames follow the
shape contract of the reference's datasets (utils/data.py:84,106,125): dict(target, reference, gt) of
float32 [3,H,W] in [0,1]; the distortion is a fixed gain/gamma/hue-like matrix so that PSNR is meaningful.
Seeded by FRAME index (world-size independent)."""
import numpy as np
import torch

from utils.sharding import frame_seed


class SyntheticStereoFrames:
    def __init__(self, n_frames=8, height=270, width=480):
        self.n_frames, self.height, self.width = int(n_frames), int(height), int(width)

    def __len__(self):
        return self.n_frames

    def __getitem__(self, f):
        rng = np.random.default_rng(frame_seed(f))
        yy, xx = np.mgrid[0:self.height, 0:self.width].astype(np.float32)
        base = np.stack([0.5 + 0.4 * np.sin(xx / self.width * 6.0 + f * 0.1) * np.cos(yy / self.height * 4.0),
                         0.2 + 0.6 * xx / self.width, 0.5 + 0.4 * np.cos((xx + yy) / (self.width + self.height) * 9.0)])
        gt = np.clip(base + 0.05 * rng.standard_normal(base.shape).astype(np.float32), 0, 1).astype(np.float32)
        reference = np.clip(np.roll(gt, 7, axis=2) + 0.02 * rng.standard_normal(base.shape).astype(np.float32), 0, 1)
        mix = np.array([[0.9, 0.08, 0.0], [0.05, 0.8, 0.05], [0.0, 0.1, 1.05]], dtype=np.float32)
        target = np.clip(np.einsum("ij,jhw->ihw", mix, gt) ** 1.15 + 0.03, 0, 1).astype(np.float32)
        return {"target": torch.from_numpy(target), "reference": torch.from_numpy(reference.astype(np.float32)),
                "gt": torch.from_numpy(gt)}


class DataModule:
    """Accepts the reference's init_args (data_dir, num_workers, crop_size, ...) and ignores what needs files."""

    def __init__(self, data_dir=None, num_workers=0, crop_size=None, image_repeats=None, batch_size=None,
                 n_frames=8, height=270, width=480, **_):
        self.dataset = SyntheticStereoFrames(n_frames, height, width)

    def test_frames(self):
        return self.dataset
