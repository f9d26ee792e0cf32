#!/opt/conda/bin/python3.9 -B
"""Generate golden vectors for methods/linear.py by running the REAL reference.

Run in the build container only (never on the GPU box):

    /opt/conda/bin/python3.9 -B tests/golden/make_golden_linear.py

Needs scikit-image (0.18.3 in /opt/conda) because the reference imports
``skimage.color`` (methods/linear.py:5).  The reference module is loaded by file
path so that ``methods/__init__.py`` (pytorch_lightning, piq) is bypassed.
Only inputs/outputs (data) are written; no reference source is copied.

Outputs: tests/golden/linear_small.npz   full arrays at 48x64 (float inputs)
         tests/golden/linear_u8_256.npz  full arrays at 256x256 (k/255 inputs, stored as uint8)
         tests/golden/linear_1080p.npz   stats + strided samples at 1080x1920 (inputs re-derived from the seed)
         tests/golden/linear_dtypes.npz  raw uint8 frames through Xiao / MK (no img_as_float there: 0..255 scale) and
                                         Reinhard with mixed float32 / float64 arguments (numpy promotion of the result)
"""
import hashlib
import importlib.util
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/methods/linear.py"
OUT = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_linear", REF)
lin = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lin)
from skimage.color import rgb2lab  # noqa: E402  (same library the reference uses)
import skimage  # noqa: E402
import scipy  # noqa: E402

META = dict(numpy=np.__version__, scipy=scipy.__version__, skimage=skimage.__version__,
            python=sys.version.split()[0])


def synth_pair(seed, h, w):
    """Bench-style synthetic pair: uniform float32 in [0,1)."""
    rng = np.random.default_rng(seed)
    t = rng.random((h, w, 3), dtype=np.float32)
    r = rng.random((h, w, 3), dtype=np.float32)
    return t, r


def graded_pair(seed, h, w):
    """A less trivial pair: smooth gradients + noise, different gamma/gain on the
    reference so the transfer actually moves colours; float32 in [0,1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([0.5 + 0.5 * np.sin(xx / w * 3.1 + 0.3) * np.cos(yy / h * 2.2),
                     (xx / w) * 0.8 + 0.1 * (yy / h),
                     0.5 + 0.45 * np.cos((xx + 2 * yy) / (w + h) * 5.0)], axis=-1)
    t = np.clip(base + 0.08 * rng.standard_normal(base.shape), 0, 1)
    r = np.clip(0.9 * np.clip(base[::-1, ::-1] + 0.05 * rng.standard_normal(base.shape), 0, 1) ** 1.4
                + np.array([0.05, 0.0, 0.08]), 0, 1)
    # include exact 0 / 1 and toe-region values (sRGB linear segment, Lab linear toe)
    t[0, 0] = (0, 0, 0)
    t[0, 1] = (1, 1, 1)
    t[0, 2] = (0.04045, 0.0404, 0.0405)
    t[0, 3] = (0.002, 0.001, 0.0)
    t[0, 4] = (1.0, 0.0, 0.0)
    t[0, 5] = (0.0, 0.0, 1.0)
    return t.astype(np.float32), r.astype(np.float32)


def run_all(t32, r32):
    """Reference results with float64 inputs (the parity target)."""
    t = t32.astype(np.float64)
    r = r32.astype(np.float64)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["reinhard"] = lin.color_transfer_between_images(t, r)
        lt = rgb2lab(t).reshape(-1, 3)
        lr = rgb2lab(r).reshape(-1, 3)
        out["lab_mean_t"], out["lab_std_t"] = lt.mean(axis=0), lt.std(axis=0)
        out["lab_mean_r"], out["lab_std_r"] = lr.mean(axis=0), lr.std(axis=0)
        # transferred image still in Lab = methods/linear.py:38 evaluated on the same arrays
        out["reinhard_lab"] = ((lt - out["lab_mean_t"]) * out["lab_std_r"] / out["lab_std_t"]
                               + out["lab_mean_r"]).reshape(t.shape)
        out["xiao"] = lin.color_transfer_in_correlated_color_space(t, r)
        for d in ("MK", "sqrt", "cholesky"):
            out["mk_" + d] = lin.monge_kantorovitch_color_transfer(t, r, decomposition=d)
        out["rgb_mean_t"], out["rgb_cov_t"] = t.reshape(-1, 3).mean(axis=0), np.cov(t.reshape(-1, 3).T)
        out["rgb_mean_r"], out["rgb_cov_r"] = r.reshape(-1, 3).mean(axis=0), np.cov(r.reshape(-1, 3).T)
        # the Runner-style float32-input call, informational (reference self-noise)
        out["reinhard_f32in"] = lin.color_transfer_between_images(t32, r32)
    return out


def main():
    # ---- small, full arrays ---------------------------------------------------------------
    small = {}
    for name, (t, r) in {"uniform": synth_pair(7, 48, 64), "graded": graded_pair(11, 48, 64)}.items():
        res = run_all(t, r)
        small[name + "/target"] = t
        small[name + "/reference"] = r
        for k, v in res.items():
            small[name + "/" + k] = v
    np.savez_compressed(os.path.join(OUT, "linear_small.npz"), meta=str(META), **small)

    # ---- 256x256 with k/255 inputs (what real frames look like: utils/data.py:84,106,125) -
    rng = np.random.default_rng(256)
    t8 = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:256, 0:256]
    r8 = np.stack([(xx * 0.7 + 30), (yy * 0.9 + 10), ((xx + yy) * 0.45 + 5)], axis=-1)
    r8 = np.clip(r8 + rng.integers(-20, 21, r8.shape), 0, 255).astype(np.uint8)
    t = t8.astype(np.float32) / 255
    r = r8.astype(np.float32) / 255
    res = run_all(t, r)
    keep = {"target_u8": t8, "reference_u8": r8}
    for k in ("lab_mean_t", "lab_std_t", "lab_mean_r", "lab_std_r", "rgb_mean_t", "rgb_cov_t",
              "rgb_mean_r", "rgb_cov_r"):
        keep[k] = res[k]
    sl = (slice(None, None, 3), slice(None, None, 3))          # every 3rd pixel both ways
    for k in ("reinhard", "reinhard_lab", "xiao", "mk_MK", "mk_sqrt", "mk_cholesky"):
        keep[k + "_s3"] = res[k][sl]
    np.savez_compressed(os.path.join(OUT, "linear_u8_256.npz"), meta=str(META), **keep)

    # ---- 1080p: stats + strided samples, inputs re-derived from the seed ---------------------
    t, r = synth_pair(1234, 1080, 1920)
    res = run_all(t, r)
    idx = np.arange(0, 1080 * 1920, 4099)
    keep = {"seed": 1234, "idx": idx,
            "target_sha256": hashlib.sha256(t.tobytes()).hexdigest(),
            "reference_sha256": hashlib.sha256(r.tobytes()).hexdigest()}
    for k in ("lab_mean_t", "lab_std_t", "lab_mean_r", "lab_std_r", "rgb_mean_t", "rgb_cov_t",
              "rgb_mean_r", "rgb_cov_r"):
        keep[k] = res[k]
    for k in ("reinhard", "reinhard_lab", "xiao", "mk_MK", "reinhard_f32in"):
        keep[k + "_samples"] = res[k].reshape(-1, 3)[idx]
    np.savez_compressed(os.path.join(OUT, "linear_1080p.npz"), meta=str(META), **keep)

    # ---- dtype rules: raw uint8 through Xiao / MK, mixed float dtypes through Reinhard -----------------------------------
    rng = np.random.default_rng(77)
    t8 = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    r8 = np.clip(rng.integers(0, 256, (24, 40, 3)) * 0.6 + 60, 0, 255).astype(np.uint8)
    keep = {"t8": t8, "r8": r8}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        keep["xiao_u8"] = lin.color_transfer_in_correlated_color_space(t8, r8)
        keep["mk_u8"] = lin.monge_kantorovitch_color_transfer(t8, r8)
        keep["reinhard_u8"] = lin.color_transfer_between_images(t8, r8)
        t32, r64 = (t8 / 255).astype(np.float32), (r8 / 255).astype(np.float64)
        keep["reinhard_f32_f64"] = lin.color_transfer_between_images(t32, r64)
        keep["reinhard_f64_f32"] = lin.color_transfer_between_images(t32.astype(np.float64), r64.astype(np.float32))
    print("dtype rules:", {k: str(v.dtype) for k, v in keep.items()}, "xiao_u8 range %.1f..%.1f" % (keep["xiao_u8"].min(), keep["xiao_u8"].max()))
    np.savez_compressed(os.path.join(OUT, "linear_dtypes.npz"), meta=str(META), **keep)
    print("wrote goldens with", META)


if __name__ == "__main__":
    main()
