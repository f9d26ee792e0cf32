"""Linear Color Transfer Methods -- MI355X drop-in for the reference's methods/linear.py.

Same names, positional signatures, kwargs, defaults, return-dtype rules and exceptions as
the reference (methods/linear.py:8,45,85): numpy HxWx3 in, numpy HxWx3 out.  The per-pixel
sweeps run as HIP kernels (libct_hip.so, include/ct_hip.h); the 3x3 algebra of Xiao / MK
stays on the host and calls the very same numpy/scipy routines as the reference
(linear.py:69-78,108-118), because LAPACK's SVD sign convention is part of the result.

Device-resident variants (`*_cuda`, torch tensors in/out, no host copies) are what
bench.py times.  There is no CPU fallback.
"""
import numpy as np
import scipy
import scipy.linalg
import torch

import ct_hip

__all__ = ["color_transfer_between_images", "color_transfer_in_correlated_color_space",
           "monge_kantorovitch_color_transfer"]


def _as_float(arr):
    """skimage.img_as_float semantics for the dtypes the reference ever sees
    (float32 from Runner, float64 from the demo, uint8 frames): floats pass through."""
    arr = np.asanyarray(arr)
    if arr.shape[-1] != 3:
        raise ValueError("Input array must have a shape == (..., 3)), got %s" % (arr.shape,))
    if arr.dtype in (np.float32, np.float64):
        return arr
    if arr.dtype == np.float16:
        return arr.astype(np.float32)
    if arr.dtype == np.uint8:
        return arr.astype(np.float64) / 255.0
    if arr.dtype == np.uint16:
        return arr.astype(np.float64) / 65535.0
    if arr.dtype == np.bool_:
        return arr.astype(np.float64)
    return arr.astype(np.float64)


def _as_raw_float(arr):
    """What Xiao / MK / IDT see in the reference (methods/linear.py:45-124, iterative.py:8-59): NO img_as_float -- only
    Reinhard goes through skimage's rgb2lab -- so integer frames keep their 0..255 (0..65535) scale; np.cov / the
    rotations upcast to float64."""
    arr = np.asanyarray(arr)
    if arr.shape[-1] != 3:
        raise ValueError("Input array must have a shape == (..., 3)), got %s" % (arr.shape,))
    return arr if arr.dtype in (np.float32, np.float64) else arr.astype(np.float64)


def _device():
    if not torch.cuda.is_available():
        raise ct_hip.CtHipError("methods.linear needs an MI355X (no CPU fallback); torch.cuda.is_available() is False")
    return torch.device("cuda", torch.cuda.current_device())


def _to_device(arr, dtype=None):
    """numpy (possibly a non-contiguous permuted view, methods/__init__.py:21-22) -> [H*W,3]-shaped
    contiguous device tensor viewed as [1, n, 1, 3]."""
    a = np.ascontiguousarray(arr, dtype=dtype)
    return torch.from_numpy(a.reshape(1, -1, 1, 3)).to(_device(), non_blocking=False)


# ------------------------------------------------------------------------------------------------
# Reinhard
# ------------------------------------------------------------------------------------------------
def color_transfer_between_images_cuda(target, reference, out=None):
    """Device-resident Reinhard transfer. target/reference: CUDA tensors [H,W,3] or [B,H,W,3]
    (float32 or float64, same dtype). Asynchronous on the current stream."""
    if target.shape == reference.shape:
        return ct_hip.reinhard(target, reference, out=out)
    st = ct_hip.lab_stats(target)
    sr = ct_hip.lab_stats(reference)
    return ct_hip.reinhard_apply(target, st, sr, out=out)


def color_transfer_between_images(target, reference):
    """Color Transfer between Images (Reinhard et al. 2001) -- reference methods/linear.py:8-42.

    Output dtype = np.result_type(target, reference) after img_as_float (skimage keeps float32; the reference's
    `(t - mean_t) * std_r / std_t + mean_r` promotes to float64 as soon as one side is float64) and is clipped to [0,1]
    (skimage xyz2rgb).
    """
    target = _as_float(target)
    reference = _as_float(reference)
    shape = target.shape
    dt = np.result_type(target.dtype, reference.dtype)
    if target.size == 0:
        return np.empty(shape, dtype=dt)
    t = _to_device(target, dt)
    r = _to_device(reference, dt)
    out = color_transfer_between_images_cuda(t, r)
    return out.cpu().numpy().reshape(shape)             # numpy promotion like the reference: float32 only if BOTH are float32


# ------------------------------------------------------------------------------------------------
# Xiao / Monge-Kantorovitch: device moments -> host 3x3 algebra -> device affine
# ------------------------------------------------------------------------------------------------
def _host_moments(t, r):
    """(mean_t, cov_t, mean_r, cov_r) as float64 numpy; one D2H copy of 2 x 16 doubles."""
    st = ct_hip.rgb_meancov(t)
    sr = ct_hip.rgb_meancov(r)
    s = torch.stack([st[0], sr[0]]).cpu().numpy()
    return s[0, 0:3], s[0, 3:12].reshape(3, 3), s[1, 0:3], s[1, 3:12].reshape(3, 3)


def _affine(t, A, mean_t, mean_r, out_dtype=torch.float64):
    coef = np.zeros((1, 16), dtype=np.float64)
    coef[0, 0:9] = np.asarray(A, dtype=np.float64).reshape(9)
    coef[0, 9:12] = mean_t
    coef[0, 12:15] = mean_r
    return ct_hip.affine3x3(t, torch.from_numpy(coef).to(t.device), out_dtype=out_dtype)


def xiao_matrix(target_cov, reference_cov):
    """T of methods/linear.py:69-78 (host, same numpy calls as the reference)."""
    target_u, target_s, _ = np.linalg.svd(target_cov)
    reference_u, reference_s, _ = np.linalg.svd(reference_cov)
    target_rotation = target_u
    reference_rotation = np.linalg.inv(reference_u)
    target_scale = np.diag(1 / np.sqrt(target_s))
    reference_scale = np.diag(np.sqrt(reference_s))
    return target_rotation @ target_scale @ reference_scale @ reference_rotation


def mk_matrix(target_cov, reference_cov, decomposition="MK"):
    """T of methods/linear.py:108-120 (host, same numpy/scipy calls as the reference)."""
    if decomposition == "cholesky":
        A = np.linalg.cholesky(target_cov)
        B = np.linalg.cholesky(reference_cov)
        T = B @ np.linalg.inv(A)
    elif decomposition == "sqrt":
        A = scipy.linalg.sqrtm(target_cov)
        B = scipy.linalg.sqrtm(reference_cov)
        T = B @ np.linalg.inv(A)
    elif decomposition == "MK":
        A = scipy.linalg.sqrtm(target_cov)
        T = np.linalg.inv(A) @ scipy.linalg.sqrtm(A @ reference_cov @ A) @ np.linalg.inv(A)
    else:
        raise ValueError("Unknown decomposition, use either 'cholesky', 'sqrt', or 'MK'")
    return T


def color_transfer_in_correlated_color_space_cuda(target, reference, out_dtype=torch.float64):
    mt, ct_, mr, cr = _host_moments(target, reference)
    T = xiao_matrix(ct_, cr)
    return _affine(target, T.T, mt, mr, out_dtype)     # reference: (x - mu) @ T.T + mu_r


def monge_kantorovitch_color_transfer_cuda(target, reference, decomposition="MK", out_dtype=torch.float64,
                                           host_algebra=False, out=None):
    """Device-resident MK transfer, [H,W,3] or [B,H,W,3] tensors.  By default the 3x3 algebra also runs on the device
    (ct_mk_coef_f64: no host synchronisation, batches of pairs per call); host_algebra=True reproduces the
    reference's scipy/numpy calls literally (single pair)."""
    if decomposition not in ("cholesky", "sqrt", "MK"):
        raise ValueError("Unknown decomposition, use either 'cholesky', 'sqrt', or 'MK'")
    if not host_algebra:
        if target.shape == reference.shape:
            return ct_hip.mk(target, reference, decomposition, out_dtype=out_dtype, out=out)
        coef = ct_hip.mk_coef(ct_hip.rgb_meancov(target), ct_hip.rgb_meancov(reference), decomposition)
        return ct_hip.affine3x3(target, coef, out_dtype=out_dtype, out=out)
    mt, ct_, mr, cr = _host_moments(target, reference)
    T = mk_matrix(ct_, cr, decomposition)
    return _affine(target, T, mt, mr, out_dtype)       # reference: (x - mu) @ T + mu_r


def color_transfer_in_correlated_color_space(target, reference):
    """Color Transfer in Correlated Color Space (Xiao & Ma 2006) -- reference methods/linear.py:45-82.
    Returns float64, unclipped (np.cov upcasts; the caller clamps, methods/__init__.py:30)."""
    target = _as_raw_float(target)
    reference = _as_raw_float(reference)
    shape = target.shape
    out = color_transfer_in_correlated_color_space_cuda(_to_device(target), _to_device(reference))
    return out.cpu().numpy().reshape(shape)


def monge_kantorovitch_color_transfer(target, reference, decomposition="MK"):
    """The Linear Monge-Kantorovitch Linear Colour Mapping (Pitie & Kokaram 2007) -- reference
    methods/linear.py:85-124.  Returns float64, unclipped."""
    if decomposition not in ("cholesky", "sqrt", "MK"):
        raise ValueError("Unknown decomposition, use either 'cholesky', 'sqrt', or 'MK'")
    target = _as_raw_float(target)
    reference = _as_raw_float(reference)
    shape = target.shape
    # the numpy drop-in keeps the reference's own LAPACK/scipy calls for the 3x3 algebra
    out = monge_kantorovitch_color_transfer_cuda(_to_device(target), _to_device(reference), decomposition, host_algebra=True)
    return out.cpu().numpy().reshape(shape)
