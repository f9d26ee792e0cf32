"""Pitie iterative distribution transfer, CPU oracle (reference methods/iterative.py:8-59).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The arithmetic lives in
``oracle/idt_oracle.c`` (scalar C, pinned evaluation order); this module is its ctypes
wrapper plus the host-side pieces (drawing rotations exactly like the reference does).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("CT_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")     # CT_ORACLE_LIB: the sanitizer build (make -C oracle asan)
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            import subprocess
            subprocess.run(["make", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_LIB_PATH)
        P = ctypes.c_void_p
        _lib.idt_oracle.restype = ctypes.c_int
        _lib.idt_oracle.argtypes = [P, ctypes.c_int64, P, ctypes.c_int64, P, P, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, P, P, P, P, P, P, P]
        _lib.idt_project.restype = None
        _lib.idt_project.argtypes = [P, ctypes.c_int64, P, P]
    return _lib


def draw_rotations(n_iter, n_dims=3):
    """Exactly what the reference does once per iteration (methods/iterative.py:32): consumes the
    GLOBAL numpy RNG through scipy, so a caller that seeds np.random gets the reference's matrices."""
    import scipy.stats
    return np.stack([scipy.stats.special_ortho_group.rvs(n_dims) for _ in range(n_iter)])


def project(x, r):
    """d = r @ x.T with the pinned FMA order; x [n,3] float64 -> [3,n]."""
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(r, dtype=np.float64)
    d = np.empty((3, x.shape[0]), dtype=np.float64)
    _load().idt_project(x.ctypes.data, x.shape[0], r.ctypes.data, d.ctypes.data)
    return d


def iterative_distribution_transfer(target, reference, bins=255, n_iter=4, rotations=None, debug=False):
    """Oracle of methods/iterative.py:8-59. `rotations` [n_iter,3,3] (default: drawn like the
    reference).  Returns float64 HxWx3; with debug=True also a dict of intermediates."""
    target = np.asarray(target)
    reference = np.asarray(reference)
    shape = target.shape
    round_f32 = 1 if target.dtype == np.float32 else 0
    t = np.ascontiguousarray(target.reshape(-1, 3), dtype=np.float64)
    r = np.ascontiguousarray(reference.reshape(-1, 3), dtype=np.float64)
    if rotations is None:
        rotations = draw_rotations(n_iter)
    rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(n_iter, 9)
    rinv = np.ascontiguousarray(np.stack([np.linalg.inv(m.reshape(3, 3)) for m in rot]).reshape(n_iter, 9))
    out = np.empty_like(t)
    n_t, n_r = t.shape[0], r.shape[0]
    dbg = {}
    ptrs = [None] * 6
    if debug:
        dbg["lohi"] = np.zeros((n_iter, 3, 2))
        dbg["hist0"] = np.zeros((n_iter, 3, bins), dtype=np.int64)
        dbg["hist1"] = np.zeros((n_iter, 3, bins), dtype=np.int64)
        dbg["lut"] = np.zeros((n_iter, 3, bins))
        dbg["binidx"] = np.zeros((n_iter, 3, n_t), dtype=np.uint16)
        dbg["state"] = np.zeros((n_iter, n_t, 3))
        ptrs = [dbg[k].ctypes.data for k in ("lohi", "hist0", "hist1", "lut", "binidx", "state")]
    rc = _load().idt_oracle(t.ctypes.data, n_t, r.ctypes.data, n_r, rot.ctypes.data, rinv.ctypes.data, n_iter, bins,
                            round_f32, out.ctypes.data, *ptrs)
    if rc != 0:
        raise RuntimeError("idt_oracle failed (%d)" % rc)
    out = out.reshape(shape)
    return (out, dbg) if debug else out
