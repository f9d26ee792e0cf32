#!/usr/bin/env python3
"""Generate golden vectors for methods/iterative.py:iterative_distribution_transfer by running the
REAL reference in the build container (never on the GPU box):

    python3 -B tests/golden/make_golden_idt.py

The reference imports `skimage.transform.resize` at module level (iterative.py:5; only used by
`_regrain`), which the system interpreter lacks, so an empty stub module is registered before the
reference module is loaded by file path.  Intermediates are captured WITHOUT modifying the
reference: `scipy.stats.special_ortho_group.rvs`, `np.histogram`, `np.bincount` and `np.interp`
are wrapped to record what flows through them (rotations; lo/hi + counts; numpy's own per-pixel
bin indices; the LUT f).  Only inputs/outputs (data) are written.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy
import scipy.stats

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/methods/iterative.py"

for name in ("skimage", "skimage.transform"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["skimage.transform"].resize = None
spec = importlib.util.spec_from_file_location("ref_iterative", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

META = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0])


class Capture:
    """Wraps numpy/scipy entry points the reference calls and records their traffic."""

    def __init__(self):
        self.rot, self.hist, self.binidx, self.interp = [], [], [], []
        self._cur_idx = None

    def __enter__(self):
        self._rvs = scipy.stats.special_ortho_group.rvs
        self._hist, self._binc, self._interp = np.histogram, np.bincount, np.interp
        cap = self

        def rvs(*a, **k):
            m = cap._rvs(*a, **k)
            cap.rot.append(np.array(m))
            return m

        def bincount(x, *a, **k):
            if cap._cur_idx is not None:
                cap._cur_idx.append(np.array(x))
            return cap._binc(x, *a, **k)

        def histogram(a, bins=10, range=None, **k):
            cap._cur_idx = []
            counts, edges = cap._hist(a, bins=bins, range=range, **k)
            cap.hist.append((np.array(range, dtype=np.float64), counts.copy()))
            cap.binidx.append(np.concatenate(cap._cur_idx) if cap._cur_idx else np.zeros(0, np.intp))
            cap._cur_idx = None
            return counts, edges

        def interp(x, xp, fp, *a, **k):
            res = cap._interp(x, xp, fp, *a, **k)
            cap.interp.append(np.array(res))
            return res

        scipy.stats.special_ortho_group.rvs = rvs
        np.histogram, np.bincount, np.interp = histogram, bincount, interp
        return self

    def __exit__(self, *exc):
        scipy.stats.special_ortho_group.rvs = self._rvs
        np.histogram, np.bincount, np.interp = self._hist, self._binc, self._interp


def run(target, reference, seed, bins=255, n_iter=4):
    np.random.seed(seed)
    with Capture() as cap:
        out = ref.iterative_distribution_transfer(target, reference, bins=bins, n_iter=n_iter)
    n_t = target.shape[0] * target.shape[1]
    rot = np.stack(cap.rot)
    lohi = np.zeros((n_iter, 3, 2))
    hist0 = np.zeros((n_iter, 3, bins), np.int64)
    hist1 = np.zeros((n_iter, 3, bins), np.int64)
    binidx = np.zeros((n_iter, 3, n_t), np.uint16)
    lut = np.zeros((n_iter, 3, bins))
    for it in range(n_iter):
        for j in range(3):
            h = (it * 3 + j) * 2               # two np.histogram calls per (iter, axis): target, reference
            lohi[it, j] = cap.hist[h][0]
            hist0[it, j] = cap.hist[h][1]
            hist1[it, j] = cap.hist[h + 1][1]
            binidx[it, j] = cap.binidx[h]
            lut[it, j] = cap.interp[(it * 3 + j) * 2]     # first np.interp of the pair builds f (255 values)
    return dict(out=out, rot=rot, lohi=lohi, hist0=hist0, hist1=hist1, binidx=binidx, lut=lut)


def graded_pair(seed, h, w, dtype):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([0.5 + 0.5 * np.sin(xx / w * 3.1 + 0.3) * np.cos(yy / h * 2.2),
                     (xx / w) * 0.8 + 0.1 * (yy / h),
                     0.5 + 0.45 * np.cos((xx + 2 * yy) / (w + h) * 5.0)], axis=-1)
    t = np.clip(base + 0.08 * rng.standard_normal(base.shape), 0, 1)
    r = np.clip(0.9 * np.clip(base[::-1, ::-1] + 0.05 * rng.standard_normal(base.shape), 0, 1) ** 1.4
                + np.array([0.05, 0.0, 0.08]), 0, 1)
    return t.astype(dtype), r.astype(dtype)


def main():
    small = {}
    # float64 inputs (demo-style) and float32 inputs (Runner-style: d_r is float32 on iteration 0)
    for name, dtype, seed in (("f64", np.float64, 3), ("f32", np.float32, 4)):
        t, r = graded_pair(21, 48, 64, dtype)
        res = run(t, r, seed)
        small[name + "/target"], small[name + "/reference"] = t, r
        for k, v in res.items():
            small[name + "/" + k] = v
    # different target/reference sizes + non-default bins/n_iter
    rng = np.random.default_rng(5)
    t = rng.random((20, 30, 3))
    r = rng.random((25, 17, 3)) * 0.7 + 0.1
    res = run(t, r, 6, bins=64, n_iter=2)
    small["odd/target"], small["odd/reference"] = t, r
    for k, v in res.items():
        small["odd/" + k] = v
    np.savez_compressed(os.path.join(OUT, "idt_small.npz"), meta=str(META), **small)

    # 256x256, k/255 float32 inputs (what Runner feeds): counts, LUTs, iteration-0 bin indices, samples
    rng = np.random.default_rng(256)
    t8 = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:256, 0:256]
    r8 = np.stack([(xx * 0.7 + 30), (yy * 0.9 + 10), ((xx + yy) * 0.45 + 5)], axis=-1)
    r8 = np.clip(r8 + rng.integers(-20, 21, r8.shape), 0, 255).astype(np.uint8)
    res = run(t8.astype(np.float32) / 255, r8.astype(np.float32) / 255, 7)
    np.savez_compressed(os.path.join(OUT, "idt_u8_256.npz"), meta=str(META), target_u8=t8, reference_u8=r8,
                        rot=res["rot"], lohi=res["lohi"], hist0=res["hist0"], hist1=res["hist1"], lut=res["lut"],
                        binidx_it0=res["binidx"][0].astype(np.uint8), out_s3=res["out"][::3, ::3])

    # 1080p synthetic (bench inputs): counts + LUT + samples
    rng = np.random.default_rng(1234)
    t = rng.random((1080, 1920, 3), dtype=np.float32)
    r = rng.random((1080, 1920, 3), dtype=np.float32)
    res = run(t, r, 0)
    idx = np.arange(0, 1080 * 1920, 4099)
    np.savez_compressed(os.path.join(OUT, "idt_1080p.npz"), meta=str(META), seed=1234, rot=res["rot"], lohi=res["lohi"],
                        hist0=res["hist0"], hist1=res["hist1"], lut=res["lut"], idx=idx,
                        out_samples=res["out"].reshape(-1, 3)[idx],
                        binidx_it0_samples=res["binidx"][0][:, idx].astype(np.uint8))
    print("wrote IDT goldens with", META)


if __name__ == "__main__":
    main()
