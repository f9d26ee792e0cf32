"""Global linear colour transfers, float64 restatement of methods/linear.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All arithmetic is float64
regardless of the input dtype: this is the reference called with
``img_as_float`` (float64) inputs, which is what the 1e-4 Lab tolerance is
defined against.
"""
import numpy as np
import scipy.linalg

from . import lab as _lab


def lab_stats(rgb):
    """Per-channel Lab mean and population std (methods/linear.py:25-36)."""
    x = _lab.rgb2lab(rgb).reshape(-1, 3)
    return np.mean(x, axis=0), np.std(x, axis=0)


def reinhard_lab(target, reference):
    """Transferred image still in Lab, i.e. methods/linear.py:25-38."""
    t = _lab.rgb2lab(target)
    r = _lab.rgb2lab(reference)
    shape = t.shape
    t = t.reshape(-1, 3)
    r = r.reshape(-1, 3)
    t_mean, r_mean = np.mean(t, axis=0), np.mean(r, axis=0)
    t_std, r_std = np.std(t, axis=0), np.std(r, axis=0)
    out = (t - t_mean) * r_std / t_std + r_mean
    return out.reshape(shape)


def color_transfer_between_images(target, reference):
    """Reinhard et al. 2001 (methods/linear.py:8-42)."""
    return _lab.lab2rgb(reinhard_lab(target, reference))


def rgb_mean_cov(img):
    """np.mean(axis=0) and np.cov(x.T) (ddof=1), methods/linear.py:64-67,103-106."""
    x = np.asarray(img, dtype=np.float64).reshape(-1, 3)
    mean = x.mean(axis=0)
    d = x - mean
    cov = d.T @ d / (x.shape[0] - 1)
    return mean, cov


def xiao_matrix(target_cov, reference_cov):
    """methods/linear.py:69-78."""
    target_u, target_s, _ = np.linalg.svd(target_cov)
    reference_u, reference_s, _ = np.linalg.svd(reference_cov)
    target_rotation = target_u
    reference_rotation = np.linalg.inv(reference_u)
    target_scale = np.diag(1 / np.sqrt(target_s))
    reference_scale = np.diag(np.sqrt(reference_s))
    return target_rotation @ target_scale @ reference_scale @ reference_rotation


def color_transfer_in_correlated_color_space(target, reference):
    """Xiao & Ma 2006 (methods/linear.py:45-82). Note ``@ T.T``."""
    shape = np.shape(target)
    t_mean, t_cov = rgb_mean_cov(target)
    r_mean, r_cov = rgb_mean_cov(reference)
    T = xiao_matrix(t_cov, r_cov)
    t = np.asarray(target, dtype=np.float64).reshape(-1, 3)
    return ((t - t_mean) @ T.T + r_mean).reshape(shape)


def mk_matrix(target_cov, reference_cov, decomposition="MK"):
    """methods/linear.py:108-120."""
    if decomposition == "cholesky":
        A = np.linalg.cholesky(target_cov)
        B = np.linalg.cholesky(reference_cov)
        return B @ np.linalg.inv(A)
    if decomposition == "sqrt":
        A = scipy.linalg.sqrtm(target_cov)
        B = scipy.linalg.sqrtm(reference_cov)
        return B @ np.linalg.inv(A)
    if decomposition == "MK":
        A = scipy.linalg.sqrtm(target_cov)
        Ainv = np.linalg.inv(A)
        return Ainv @ scipy.linalg.sqrtm(A @ reference_cov @ A) @ Ainv
    raise ValueError("Unknown decomposition, use either 'cholesky', 'sqrt', or 'MK'")


def monge_kantorovitch_color_transfer(target, reference, decomposition="MK"):
    """Pitie & Kokaram 2007 (methods/linear.py:85-124). Note ``@ T`` (no transpose)."""
    shape = np.shape(target)
    t_mean, t_cov = rgb_mean_cov(target)
    r_mean, r_cov = rgb_mean_cov(reference)
    T = mk_matrix(t_cov, r_cov, decomposition)
    t = np.asarray(target, dtype=np.float64).reshape(-1, 3)
    return ((t - t_mean) @ T + r_mean).reshape(shape)
