"""Iterative Color Transfer Methods -- MI355X drop-in for the reference's methods/iterative.py.

`iterative_distribution_transfer(target, reference, bins=255, n_iter=4)` keeps the reference's
signature, defaults and return rule (float64 HxWx3, unclipped; methods/iterative.py:8-59).  The
rotations are drawn on the host exactly like the reference does -- one
`scipy.stats.special_ortho_group.rvs(3)` per iteration from numpy's GLOBAL RNG
(iterative.py:32) -- so a caller that seeds `np.random` gets the same matrices; everything per
pixel runs in HIP kernels (libct_hip.so: ct_idt_f32 / ct_idt_f64).  Extra keyword `rotations=`
lets a caller pass the matrices explicitly (used by the parity tests and by world-size
independent frame sharding).  No CPU fallback.

`automated_color_grading(target, reference)` (iterative.py:118-138) = IDT followed by `_regrain` (iterative.py:62-117:
multigrid gradient-preserving relaxation over a skimage.transform.resize pyramid), both on the device
(ct_idt_* + ct_regrain_f64); returns float64 like the reference.
"""
import numpy as np
import scipy
import scipy.stats
import torch

import ct_hip
from methods.linear import _as_raw_float, _device

__all__ = ["iterative_distribution_transfer", "automated_color_grading", "draw_rotations"]


def draw_rotations(n_iter, n_dims=3, seed=None):
    """n_iter Haar rotations drawn like the reference (methods/iterative.py:32). `seed` (optional)
    seeds numpy's global RNG first, for callers that want frame-indexed determinism."""
    if seed is not None:
        np.random.seed(seed)
    if n_iter == 0:
        return np.zeros((0, n_dims, n_dims))
    return np.stack([scipy.stats.special_ortho_group.rvs(n_dims) for _ in range(n_iter)])


def iterative_distribution_transfer_cuda(target, reference, bins=255, n_iter=4, rotations=None, out=None):
    """Device-resident IDT: CUDA tensors [H,W,3] / [B,H,W,3] in, float64 tensor out (asynchronous)."""
    if rotations is None:
        rotations = draw_rotations(n_iter)
    if tuple(rotations.shape)[-3] == 0:             # n_iter = 0: the reference's loop body never runs
        res = target.double()
        if out is not None:
            out.copy_(res)
            return out
        return res
    return ct_hip.idt(target, reference, rotations, bins=bins, out=out)


def iterative_distribution_transfer(target, reference, bins=255, n_iter=4, rotations=None):
    """Iterative Distribution Transfer (Pitie et al. 2007) -- reference methods/iterative.py:8-59."""
    target = _as_raw_float(target)                  # no rescaling: integer frames keep their scale (iterative.py:31-55)
    reference = _as_raw_float(reference)
    shape = target.shape
    if shape[-1] != 3:
        raise ValueError("only 3-channel images are supported (n_dims = 3)")
    if rotations is None:
        rotations = draw_rotations(n_iter)          # consumes the global RNG even when nothing else runs
    rotations = np.asarray(rotations, dtype=np.float64)
    n_iter = rotations.shape[0]
    if n_iter == 0:
        return target                               # the reference's loop body never runs
    if target.size == 0:
        return np.empty(shape, dtype=np.float64)
    if reference.size == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    dt = np.result_type(target.dtype, reference.dtype)
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(target, dtype=dt).reshape(1, -1, 1, 3)).to(dev)
    r = torch.from_numpy(np.ascontiguousarray(reference, dtype=dt).reshape(1, -1, 1, 3)).to(dev)
    # d_r is float32 on iteration 0 only when the TARGET the caller passed was float32 (iterative.py:36)
    out = ct_hip.idt(t, r, rotations, bins=bins, round_dr_f32=(target.dtype == np.float32))
    return out.cpu().numpy().reshape(shape)


def automated_color_grading_cuda(target, reference, rotations=None, nbits=(4, 16, 32, 64, 64, 64)):
    """Device-resident automated_color_grading: CUDA tensors [H,W,3] in, float64 [H,W,3] out (asynchronous)."""
    if target.dim() != 3:
        raise ValueError("automated_color_grading_cuda takes one [H,W,3] frame per call")
    graded = iterative_distribution_transfer_cuda(target, reference, rotations=rotations)
    return ct_hip.regrain(target, graded, nbits)


def automated_color_grading(target, reference, rotations=None):
    """Automated Colour Grading using Colour Distribution Transfer (Pitie et al. 2007) -- reference
    methods/iterative.py:118-138: `_regrain(target, iterative_distribution_transfer(target, reference))`."""
    target = _as_raw_float(target)
    reference = _as_raw_float(reference)
    if target.ndim != 3 or target.shape[-1] != 3:
        raise ValueError("not enough values to unpack (expected 3)" if target.ndim < 3 else "only H x W x 3 images are supported")
    graded = iterative_distribution_transfer(target, reference, rotations=rotations)
    if target.size == 0:
        return graded
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(target, dtype=np.float64)).to(dev)
    out = ct_hip.regrain(t, torch.from_numpy(np.ascontiguousarray(graded)).to(dev))
    return out.cpu().numpy()
