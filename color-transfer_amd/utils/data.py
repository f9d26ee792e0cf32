"""Test-time data path of the reference (utils/data.py:12-22,87-125,168-179) for this stack.

* `setup_grid_distortions` -- the 31 deterministic distortions of `ArtificialTestDataset` (identity + brightness / contrast /
  saturation / hue / gamma at 6 magnitudes), as (name, parameter) pairs; they are APPLIED on the GPU (ct_distort_u8:
  torchvision's uint8 tensor arithmetic restated) to the uint8 ground-truth frame, index -> (image index // 31, distortion
  index % 31) exactly like the reference.
* `ArtificialTestDataset(image_dir)` / `RealWorldTestDataset(image_dir)` -- the reference's file layouts (`*_L.*`, `*_R.*`,
  `*/*_LD.*`); PNG decoding stays on the host (PIL here, libpng behind torchvision.io.read_image there).
* `SyntheticArtificialTest` / `SyntheticStereoFrames` -- stand-ins when the Kaggle / MSU datasets are not on disk (they are not
  available offline): synthetic uint8 stereo pairs through the same 31 distortions, or a synthetic float video.
* `prefetch(dataset, indices, device)` -- yields device-resident samples while the next one is decoded into PINNED host
  memory and uploaded on a second stream (double buffer), so that the transfer kernels never wait for PCIe.
Seeds are derived from the FRAME index (world-size independent)."""
import os
from pathlib import Path

import numpy as np
import torch

from utils.sharding import frame_seed


def setup_grid_distortions(max_magnitude=0.5, num=6):
    """utils/data.py:12-22: identity, then per magnitude in linspace(-m, m, num): brightness, contrast, saturation (factor
    1 + magnitude), hue (factor = magnitude), gamma (1 + magnitude)."""
    specs = [("identity", 0.0)]
    for magnitude in np.linspace(-max_magnitude, max_magnitude, num):
        specs += [("brightness", 1 + magnitude), ("contrast", 1 + magnitude), ("saturation", 1 + magnitude),
                  ("hue", magnitude), ("gamma", 1 + magnitude)]
    return specs


def read_image(path):
    """uint8 [3,H,W] like torchvision.io.read_image (RGB)"""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


class _Distorted:
    """common part of the artificial test sets: sample index -> (image, distortion); uint8 host frames in, device floats out"""

    def __init__(self):
        self.distortions = setup_grid_distortions()

    def __len__(self):
        return self.n_images() * len(self.distortions)

    def host_frames(self, index):
        """uint8 CHW (gt, reference) of sample `index` on the host + the distortion to apply to gt"""
        gt, reference = self.load_pair(index // len(self.distortions))
        return {"gt": gt, "reference": reference}, self.distortions[index % len(self.distortions)]

    @staticmethod
    def finish(dev_u8, distortion):
        """device uint8 frames -> the reference's sample dict (utils/data.py:125): gt / 255, reference / 255, target / 255"""
        import ct_hip
        out = {k: ct_hip.distort_u8(v.contiguous(), "identity", 0.0) for k, v in dev_u8.items()}      # x / 255 in float32
        out["target"] = ct_hip.distort_u8(dev_u8["gt"].contiguous(), distortion[0], distortion[1])
        return out

    def __getitem__(self, index):
        host, d = self.host_frames(index)
        return self.finish({k: v.cuda() for k, v in host.items()}, d)


class ArtificialTestDataset(_Distorted):                      # utils/data.py:107-125
    def __init__(self, image_dir):
        super().__init__()
        image_dir = Path(image_dir)
        self.gts = sorted(image_dir.glob("*_L.*"))
        self.references = sorted(image_dir.glob("*_R.*"))
        assert len(self.gts) == len(self.references)

    def n_images(self):
        return len(self.gts)

    def load_pair(self, i):
        return read_image(str(self.gts[i])), read_image(str(self.references[i]))


class SyntheticArtificialTest(_Distorted):
    """synthetic uint8 stereo pairs (a textured left view and a shifted right view) through the 31 test distortions"""

    def __init__(self, n_images=2, height=270, width=480):
        super().__init__()
        self._n, self.height, self.width = int(n_images), int(height), int(width)
        self._cache = {}

    def n_images(self):
        return self._n

    def load_pair(self, i):
        if i not in self._cache:
            rng = np.random.default_rng(frame_seed(i))
            coarse = rng.integers(0, 256, (3, self.height // 16 + 2, self.width // 16 + 2)).astype(np.float32)
            up = torch.nn.functional.interpolate(torch.from_numpy(coarse)[None], size=(self.height, self.width + 16), mode="bilinear",
                                                 align_corners=True)[0]
            up = (up + torch.from_numpy(rng.integers(-12, 13, up.shape).astype(np.float32))).clamp(0, 255).to(torch.uint8)
            self._cache = {i: (up[:, :, :self.width].contiguous(), up[:, :, 16:].contiguous())}        # keep one pair
        return self._cache[i]


class RealWorldTestDataset:                                   # utils/data.py:128-145
    def __init__(self, image_dir):
        image_dir = Path(image_dir)
        self.gts = sorted(image_dir.glob("*/*_L.*"))
        self.targets = sorted(image_dir.glob("*/*_LD.*"))
        self.references = sorted(image_dir.glob("*/*_R.*"))
        assert len(self.gts) == len(self.targets) == len(self.references)

    def __len__(self):
        return len(self.gts)

    def host_frames(self, index):
        return {"gt": read_image(str(self.gts[index])), "reference": read_image(str(self.references[index])),
                "target": read_image(str(self.targets[index]))}, None

    @staticmethod
    def finish(dev_u8, _):
        import ct_hip
        return {k: ct_hip.distort_u8(v.contiguous(), "identity", 0.0) for k, v in dev_u8.items()}

    def __getitem__(self, index):
        host, _ = self.host_frames(index)
        return self.finish({k: v.cuda() for k, v in host.items()}, None)


class SyntheticStereoFrames:
    """a synthetic float video: dict(target, reference, gt) of float32 [3,H,W] in [0,1]; the distortion is a fixed
    gain / gamma / hue-like matrix so that PSNR is meaningful"""

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


def prefetch(dataset, indices, device):
    """Yield (index, sample) with every tensor of `sample` resident on `device` as [3,H,W] float32.  Datasets with
    `host_frames` (uint8 files / synthetic uint8) go through two pinned staging buffers and a copy stream: while the caller
    works on sample i, sample i+1 is decoded on the host and uploaded; the uint8 -> float conversion and the distortion run
    on the GPU.  Other datasets (host float tensors) are uploaded the same way, converted nowhere."""
    indices = list(indices)
    if not indices:
        return
    if torch.device(device).type == "cpu":          # utils.cli's CT_CLI_DEVICE=cpu test mode: host float datasets as they are
        for index in indices:
            yield index, dataset[index]
        return
    copy_stream = torch.cuda.Stream(device=device)
    staged = [None, None]
    pinned = [{}, {}]
    slot_done = [None, None]                        # event of the last upload issued from each slot's pinned buffers

    def stage(slot, index):
        if hasattr(dataset, "host_frames"):
            host, extra = dataset.host_frames(index)
        else:
            host, extra = dataset[index], "float"
        dev = {}
        # the pinned buffers of this slot are the SOURCE of the asynchronous copy issued two samples ago: the host must not
        # refill them before that DMA has finished (the consumer stream waits on the event, the host so far did not)
        if slot_done[slot] is not None:
            slot_done[slot].synchronize()
        with torch.cuda.stream(copy_stream):
            for k, v in host.items():
                buf = pinned[slot].get(k)
                if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                    buf = pinned[slot][k] = torch.empty(v.shape, dtype=v.dtype).pin_memory()
                buf.copy_(v)
                dev[k] = buf.to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        slot_done[slot] = ev
        staged[slot] = (index, dev, extra, ev)

    stage(0, indices[0])
    for n, _ in enumerate(indices):
        slot = n & 1
        if n + 1 < len(indices):
            stage(slot ^ 1, indices[n + 1])          # decode + upload the next sample while this one is processed
        index, dev, extra, ev = staged[slot]
        torch.cuda.current_stream(device).wait_event(ev)
        for v in dev.values():
            v.record_stream(torch.cuda.current_stream(device))
        yield index, (dev if extra == "float" else dataset.finish(dev, extra))


class SyntheticStereoVideoU8:
    """BASELINE.json configs[4]: a synthetic 1080p stereo video of uint8 frames -- what every real frame is (the reference's
    datasets decode to uint8 and divide by 255, utils/data.py:84,106,125) -- handed over as PINNED HOST chunks of `group`
    consecutive frames per rank (default 8), [3 roles: target, reference, gt][group][H][W][3].  Decoding / generating 1000 distinct frames on
    the host would measure the host, so the frames cycle through a pool of `pool` distinct chunks (seeded 4321 + i)."""
    roles = ("target", "reference", "gt")

    def __init__(self, n_frames=1000, height=1080, width=1920, group=8, pool=8):
        self.n_frames, self.height, self.width, self.group, self.pool = int(n_frames), int(height), int(width), int(group), int(pool)
        self._chunks = None

    def __len__(self):
        return self.n_frames

    def _chunk(self, i):
        """pinned chunk i of the pool, generated on first use (a run that touches two chunks pins two: 149 MB each at 1080p)"""
        if self._chunks is None:
            self._chunks = [None] * self.pool
        if self._chunks[i] is None:
            rng = np.random.default_rng(4321 + i)
            trip = rng.integers(0, 256, (3, self.group, self.height, self.width, 3), dtype=np.uint8)
            t = torch.from_numpy(trip)
            self._chunks[i] = t.pin_memory() if torch.cuda.is_available() else t
        return self._chunks[i]

    def _pool_index(self, first_frame):
        """which pool chunk a group that starts at `first_frame` takes: a multiplicative hash of the frame index, so that the pool
        cycles whatever the stride between a rank's groups is (round 5 indexed by first_frame % pool: with group = pool = 8 every
        group of a rank got the same chunk) -- a function of the frame index only, not of rank or world size"""
        return ((int(first_frame) * 2654435761) >> 9) % self.pool

    def prepare(self, indices):
        """generate (and pin) every pool chunk the groups of `indices` will take -- a caller that times its loop does this first:
        generating frames on the host would measure the host (class docstring)"""
        idx = list(indices)
        for c in range(0, len(idx), self.group):
            self._chunk(self._pool_index(idx[c]))

    def host_chunk(self, first_frame):
        """the pinned chunk of the group of frames that starts at `first_frame`.  Synthetic data: a group's CONTENT is defined by its
        first frame (chunk _pool_index(first_frame), slots 0..group-1 in order), so per-frame tables depend on how frames are
        grouped (group size, world size); throughput does not."""
        return self._chunk(self._pool_index(first_frame))

    def __getitem__(self, f):
        """one frame as the reference's sample dict (float32 CHW in [0,1]): the per-sample path of utils.cli = a group of one
        (slot 0 of the chunk host_chunk(f) would hand over)"""
        c = self.host_chunk(f)[:, 0]
        return {k: (c[i].permute(2, 0, 1).float() / 255) for i, k in enumerate(self.roles)}


def prefetch_groups(dataset, indices, device, depth=3):
    """Yield (frame indices of the group, device uint8 tensor [3, k, H, W, 3]) for a dataset with `host_chunk` (pinned uint8 chunks
    of `dataset.group` frames): the uploads run on a copy stream, depth - 1 uploads ahead of the consumer, through a ring of
    `depth` device buffers; a buffer is overwritten only after the kernels that read its previous chunk are done (the consumer's
    stream is the current one when the generator is advanced).  Copy size is what the PCIe rate hangs on (measured round 5,
    1080p triples, same box): one triple per copy 47.8 GB/s, four (75 MB) 49-52, eight (149 MB, the default group) 56."""
    indices = list(indices)
    g = dataset.group
    n_chunks = (len(indices) + g - 1) // g
    if n_chunks == 0:
        return
    copy_stream = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)
    first = dataset.host_chunk(indices[0])
    ring = [torch.empty(first.shape, dtype=first.dtype, device=device) for _ in range(depth)]
    uploaded = [torch.cuda.Event() for _ in range(depth)]
    consumed = [torch.cuda.Event() for _ in range(depth)]

    def count(c):
        return min(g, len(indices) - c * g)

    def upload(c):
        slot, k = c % depth, count(c)
        with torch.cuda.stream(copy_stream):
            if c >= depth:
                copy_stream.wait_event(consumed[slot])
            src = dataset.host_chunk(indices[c * g])
            if k == g:
                ring[slot].copy_(src, non_blocking=True)
            else:                                               # ragged last chunk: only the frames that exist
                ring[slot][:, :k].copy_(src[:, :k], non_blocking=True)
            uploaded[slot].record(copy_stream)

    for c in range(min(depth - 1, n_chunks)):
        upload(c)
    for c in range(n_chunks):
        if c + depth - 1 < n_chunks:
            upload(c + depth - 1)
        slot, k = c % depth, count(c)
        main.wait_event(uploaded[slot])
        yield indices[c * g:c * g + k], ring[slot][:, :k]
        consumed[slot].record(main)


class DataModule:
    """Accepts the reference's init_args (data_dir, num_workers, crop_size, ...).  `test_frames()` = the reference's first test
    loader (ArtificialTestDataset over data_dir / "Test") when that directory exists; otherwise a synthetic stand-in:
    `synthetic: video` (default, n_frames float frames), `synthetic: artificial` (n_frames uint8 pairs x 31 distortions) or
    `synthetic: video_u8` (n_frames uint8 frames in pinned groups of `group` = 8: configs[4])."""

    def __init__(self, data_dir=None, num_workers=0, crop_size=None, image_repeats=None, batch_size=None,
                 n_frames=8, height=270, width=480, synthetic="video", group=8, **_):
        self.data_dir = Path(data_dir) if data_dir else None
        if self.data_dir is not None and (self.data_dir / "Test").is_dir():
            self.dataset = ArtificialTestDataset(self.data_dir / "Test")
        elif synthetic == "artificial":
            self.dataset = SyntheticArtificialTest(n_frames, height, width)
        elif synthetic == "video_u8":
            self.dataset = SyntheticStereoVideoU8(n_frames, height, width, group=group)
        else:
            self.dataset = SyntheticStereoFrames(n_frames, height, width)

    def test_frames(self):
        return self.dataset

    def test_dataloader(self):
        """[artificial, real-world] like utils/data.py:168-179 (the second only when its directory exists)"""
        loaders = [self.dataset]
        if self.data_dir is not None and (self.data_dir / "Real-World Test").is_dir():
            loaders.append(RealWorldTestDataset(self.data_dir / "Real-World Test"))
        return loaders
