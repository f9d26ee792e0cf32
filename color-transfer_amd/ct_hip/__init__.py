"""ctypes binding of libct_hip.so (C ABI: include/ct_hip.h) + thin torch-tensor helpers.

PyTorch is used here only as the device allocator / stream provider: every function
takes CUDA (= HIP on ROCm) tensors, passes raw device pointers and the current HIP stream
to the library and returns without synchronising.

There is NO CPU fallback: importing works anywhere (so that `-m "not gpu"` tests can
check the exported symbols), but the first compute call without the library or without a
GPU raises.
"""
import ctypes
import os
import threading
import numpy as np

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CT_HIP_LIB") or os.path.join(_HERE, "libct_hip.so")   # CT_HIP_LIB: tuning builds only

CT_ABI_VERSION = 9            # include/ct_hip.h: CT_ABI_VERSION; lib() refuses any other library
CT_LAB_STATS_STRIDE = 8
CT_RGB_STATS_STRIDE = 16
CT_WS_LAB_STATS, CT_WS_RGB_MEANCOV, CT_WS_REINHARD, CT_WS_IDT, CT_WS_REINHARD_PSNR, CT_WS_REINHARD_PERSIST = 0, 1, 2, 3, 4, 5

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_p = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check it against include/ct_hip.h
SIGNATURES = {
    "ct_abi_version": (_c_int, []),
    "ct_error_string": (ctypes.c_char_p, [_c_int]),
    "ct_profile_events": (None, [_c_p, _c_p, _c_p, _c_p]),
    "ct_set_lab_mode": (_c_int, [_c_int]),
    "ct_get_lab_mode": (_c_int, []),
    "ct_set_lab_mode_thread": (_c_int, [_c_int]),
    "ct_workspace_bytes": (_c_sz, [_c_int, _c_i64, _c_int]),
    "ct_lab_stats_f32": (_c_int, [_c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_lab_stats_f64": (_c_int, [_c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_reinhard_apply_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_reinhard_apply_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_reinhard_lab_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_reinhard_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_reinhard_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_reinhard_psnr_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_device_status": (_c_int, [_c_int]),
    "ct_reinhard_persist_supported": (_c_int, [_c_i64]),
    "ct_reinhard_takes_persist": (_c_int, [_c_i64]),
    "ct_reinhard_persist_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_reinhard_psnr_u8": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_rgb_meancov_f32": (_c_int, [_c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_rgb_meancov_f64": (_c_int, [_c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_mk_f32_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_int, _c_p, _c_sz, _c_p]),
    "ct_mk_f32_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_int, _c_p, _c_sz, _c_p]),
    "ct_mk_f64_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_int, _c_p, _c_sz, _c_p]),
    "ct_frame_psnr_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_distort_u8": (_c_int, [_c_p, _c_int, _c_int, _c_int, ctypes.c_double, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "ct_regrain_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "ct_regrain_f64": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p, _c_int, _c_p, _c_sz, _c_p]),
    "ct_metric_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "ct_frame_ssim_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_frame_icid_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_sz, _c_p]),
    "ct_mk_coef_f64": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_p, _c_p]),
    "ct_affine3x3_f32_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_affine3x3_f64_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_affine3x3_f32_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "ct_idt_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "ct_idt_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_sz,
                            _c_p, _c_p]),
    "ct_idt_f64": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_sz,
                            _c_p, _c_p]),
}


class IdtDebug(ctypes.Structure):
    """struct ct_idt_debug (include/ct_hip.h)"""
    _fields_ = [("hist", _c_p), ("lut", _c_p), ("par", _c_p), ("binidx", _c_p)]

_lib = None
_lock = threading.RLock()          # re-entrant: lib() takes it on first use, possibly under a caller that already holds it


class CtHipError(RuntimeError):
    pass


def lib():
    """Load libct_hip.so (once). Raises loudly when it is missing -- there is no fallback."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise CtHipError(
                        "HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C color-transfer_amd/csrc`; this package has no CPU fallback" % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)  # AttributeError = ABI mismatch, also loud
                    fn.restype = res
                    fn.argtypes = args
                got = handle.ct_abi_version()
                if got != CT_ABI_VERSION:            # a stale build (or CT_HIP_LIB) would misread every changed argument list
                    raise CtHipError("%s reports ABI version %d, this binding needs %d: rebuild with `make -C color-transfer_amd/csrc`"
                                     % (LIB_PATH, got, CT_ABI_VERSION))
                _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise CtHipError("libct_hip: %s (code %d)" % (lib().ct_error_string(rc).decode(), rc))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_device(t):
    """Kernels launch on the CURRENT device's stream: a tensor that lives elsewhere would be touched from the wrong
    context.  One process drives one GPU here, so this is an error, not a device switch."""
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise CtHipError("tensor on %s but the current device is cuda:%d; call torch.cuda.set_device(%d) first"
                         % (t.device, torch.cuda.current_device(), t.device.index))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise CtHipError("ct_hip needs device tensors (got %s); no CPU path exists" % t.device)
        if not t.is_contiguous():
            raise CtHipError("ct_hip needs contiguous HWC tensors")
        _check_device(t)


_ws_cache = {}


def workspace(kind, n_pixels, n_images, device, need=None):
    """Per-(device, stream) scratch buffer, grown on demand (never shrinks)."""
    if need is None:
        need = lib().ct_workspace_bytes(kind, n_pixels, n_images)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    with _lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
            _ws_cache[key] = buf
    return buf


_upload_rings = {}                               # device index -> [pinned uint8 [slots, bytes], events, next slot]
_UPLOAD_SLOTS, _UPLOAD_BYTES = 32, 1 << 14


def _upload_small(arr, device):
    """A small host array -> device tensor WITHOUT a blocking copy: staged through a ring of pinned slots and copied
    asynchronously on the current stream (a pageable `.to(device)` is a synchronous hipMemcpy: it would drain the stream on
    every call and serialise the host with the GPU)."""
    arr = np.ascontiguousarray(arr)
    if arr.nbytes > _UPLOAD_BYTES:
        return torch.from_numpy(arr).to(device)
    with _lock:
        ring = _upload_rings.get(device.index)
        if ring is None:
            ring = [torch.empty((_UPLOAD_SLOTS, _UPLOAD_BYTES), dtype=torch.uint8).pin_memory(), [None] * _UPLOAD_SLOTS, 0]
            _upload_rings[device.index] = ring
        slot = ring[2]
        ring[2] = (slot + 1) % _UPLOAD_SLOTS
    if ring[1][slot] is not None:
        ring[1][slot].synchronize()                # the copy that last used this slot (32 calls ago) has long finished
    host = ring[0][slot, :arr.nbytes].view(torch.from_numpy(arr).dtype).view(arr.shape)
    host.copy_(torch.from_numpy(arr))
    dev = torch.empty(arr.shape, dtype=host.dtype, device=device)
    dev.copy_(host, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ring[1][slot] = ev
    return dev


def _as_batch(img):
    """[H,W,3] or [B,H,W,3] -> ([B,H,W,3] view, had_batch_dim)."""
    if img.dim() == 3:
        return img.unsqueeze(0), False
    if img.dim() == 4:
        return img, True
    raise CtHipError("expected [H,W,3] or [B,H,W,3], got %s" % (tuple(img.shape),))


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise CtHipError("unsupported dtype %s (float32/float64 only)" % t.dtype)


CT_LAB_TABLE, CT_LAB_EXACT = 0, 1


def set_lab_mode(mode, thread=False):
    """Lab arithmetic of the float32 Reinhard entries: "table" (default; LDS look-up tables, Lab within ~5e-7 of the
    float64 path) or "exact" (float64 with hardware seeds).  Process-wide default (ct_set_lab_mode), or -- thread=True -- an
    override for the calling thread only (ct_set_lab_mode_thread; mode None removes it)."""
    if thread and mode is None:
        check(lib().ct_set_lab_mode_thread(-1))
        return
    code = {"table": CT_LAB_TABLE, "exact": CT_LAB_EXACT}.get(mode)
    if code is None:
        raise ValueError("lab mode must be 'table' or 'exact', got %r" % (mode,))
    check(lib().ct_set_lab_mode_thread(code) if thread else lib().ct_set_lab_mode(code))


def lab_mode():
    return "exact" if lib().ct_get_lab_mode() == CT_LAB_EXACT else "table"


def profile_events(events):
    """events: None (off) or four torch.cuda.Event(enable_timing=True) that have been recorded once (so that their
    hipEvent_t exists); the library re-records them around moments_kernel / reinhard_apply_kernel."""
    if events is None:
        lib().ct_profile_events(None, None, None, None)
    else:
        lib().ct_profile_events(*[ctypes.c_void_p(e.cuda_event) for e in events])


def lab_stats(img):
    """rgb2lab + mean/std (population) per image: returns float64 [B, 8] = mean[3], std[3], n, 0.
    Replaces methods/linear.py:25-26,33-36."""
    x, _ = _as_batch(img)
    _require_cuda(x)
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    stats = torch.empty((B, CT_LAB_STATS_STRIDE), dtype=torch.float64, device=x.device)
    ws = workspace(CT_WS_LAB_STATS, n, B, x.device)
    fn = getattr(lib(), "ct_lab_stats_" + _suffix(x))
    check(fn(_ptr(x), n, B, _ptr(stats), _ptr(ws), ws.numel(), _stream()))
    return stats


def rgb_meancov(img):
    """np.mean + np.cov (ddof 1) per image: float64 [B, 16] = mean[3], cov[9], n, 0,0,0.
    Replaces methods/linear.py:64-67,103-106."""
    x, _ = _as_batch(img)
    _require_cuda(x)
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    stats = torch.empty((B, CT_RGB_STATS_STRIDE), dtype=torch.float64, device=x.device)
    ws = workspace(CT_WS_RGB_MEANCOV, n, B, x.device)
    fn = getattr(lib(), "ct_rgb_meancov_" + _suffix(x))
    check(fn(_ptr(x), n, B, _ptr(stats), _ptr(ws), ws.numel(), _stream()))
    return stats


def reinhard_apply(target, stats_t, stats_r, out=None, to_lab=False):
    """Affine map in Lab + lab2rgb (methods/linear.py:38-40); stats stay on the device."""
    x, _ = _as_batch(target)
    _require_cuda(x, stats_t, stats_r)
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if out is None:
        out = torch.empty_like(x)
    if to_lab:
        if x.dtype != torch.float32:
            raise CtHipError("the Lab probe exists for float32 only")
        fn = lib().ct_reinhard_lab_f32
    else:
        fn = getattr(lib(), "ct_reinhard_apply_" + _suffix(x))
    check(fn(_ptr(x), _ptr(stats_t), _ptr(stats_r), _ptr(out), n, B, _stream()))
    return out.view(target.shape)


def reinhard(target, reference, out=None, stats_out=None):
    """methods.linear.color_transfer_between_images on device tensors, B pairs per call
    (same H x W for target and reference).  One stats sweep over all 2B images, a finishing
    kernel, one apply sweep; no host synchronisation."""
    x, _ = _as_batch(target)
    r, _ = _as_batch(reference)
    _require_cuda(x, r)
    if x.shape != r.shape or x.dtype != r.dtype:
        raise CtHipError("fused reinhard needs equal shapes/dtypes; use lab_stats + reinhard_apply otherwise")
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if out is None:
        out = torch.empty_like(x)
    ws = workspace(CT_WS_REINHARD, n, B, x.device)
    fn = getattr(lib(), "ct_reinhard_" + _suffix(x))
    if stats_out is not None:
        _require_cuda(stats_out)
        if stats_out.dtype != torch.float64 or stats_out.numel() < 2 * B * CT_LAB_STATS_STRIDE:
            raise CtHipError("stats_out must be float64 with >= 2*B*8 elements")
    sp = _ptr(stats_out) if stats_out is not None else ctypes.c_void_p(0)
    check(fn(_ptr(x), _ptr(r), _ptr(out), n, B, sp, _ptr(ws), ws.numel(), _stream()))
    return out.view(target.shape)


def reinhard_psnr(target, reference, gt, out=None, psnr_out=None):
    """color_transfer_between_images for B float32 pairs + the per-frame PSNR of the result against `gt` (same layout as the
    images), as Runner.test_step computes it (methods/__init__.py:30-32).  Returns (out, psnr float64 [B, 2] = mse, PSNR)."""
    x, _ = _as_batch(target)
    r, _ = _as_batch(reference)
    g, _ = _as_batch(gt)
    _require_cuda(x, r, g)
    if not (x.shape == r.shape == g.shape) or not (x.dtype == r.dtype == g.dtype == torch.float32):
        raise CtHipError("reinhard_psnr needs three float32 tensors of one shape")
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if out is None:
        out = torch.empty_like(x)
    if psnr_out is None:
        psnr_out = torch.empty((B, 2), dtype=torch.float64, device=x.device)
    ws = workspace(CT_WS_REINHARD_PSNR, n, B, x.device)
    check(lib().ct_reinhard_psnr_f32(_ptr(x), _ptr(r), _ptr(g), _ptr(out), _ptr(psnr_out), n, B, ctypes.c_void_p(0), _ptr(ws), ws.numel(),
                                     _stream()))
    return out.view(target.shape), psnr_out


def device_status(clear=False, sync=True):
    """Sticky status bits of the current device (include/ct_hip.h: ct_device_status): 0 = all well; bit 0 = a persistent Reinhard
    launch gave up a bounded spin (its frames and PSNR records are NaN), bit 1 = a stream-K convolution gave up.  For loops that
    never synchronise per call: check once at the end (utils/sharding.gather_frame_metrics does)."""
    if sync:
        torch.cuda.synchronize()
    v = lib().ct_device_status(1 if clear else 0)
    if v < 0:
        raise CtHipError("ct_device_status: the device could not be read")
    return v


def reinhard_persist_supported(n_pixels):
    """True when frames of n_pixels can take the one-launch Reinhard kernel on this device (csrc/reinhard_persist.hip)."""
    return bool(lib().ct_reinhard_persist_supported(int(n_pixels)))


def reinhard_takes_persist(n_pixels):
    """True when reinhard() / reinhard_psnr() run float32 frames of n_pixels as the persistent launch (current Lab mode)."""
    return bool(lib().ct_reinhard_takes_persist(int(n_pixels)))


def reinhard_persist(target, reference, gt=None, out=None, psnr_out=None, stats_out=None, verify=False):
    """color_transfer_between_images (methods/linear.py:8-42) for B pairs as ONE persistent launch, optionally with the
    per-frame PSNR against `gt`.  float32 frames in [0,1] or uint8 frames (the reference's `.float() / 255`, utils/data.py:84);
    the result is float32.  Returns out, or (out, psnr [B, 2]) with gt.  verify=True synchronises and raises if a workgroup of
    the grid never became resident (results NaN)."""
    x, _ = _as_batch(target)
    r, _ = _as_batch(reference)
    ts = [x, r]
    g = None
    if gt is not None:
        g, _ = _as_batch(gt)
        ts.append(g)
    _require_cuda(*ts)
    if any(t.shape != x.shape or t.dtype != x.dtype for t in ts) or x.dtype not in (torch.float32, torch.uint8):
        raise CtHipError("reinhard_persist needs float32 or uint8 tensors of one shape")
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if not reinhard_persist_supported(n):
        raise CtHipError("frames of %d pixels do not fit the persistent Reinhard launch on this device" % n)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if g is not None and psnr_out is None:
        psnr_out = torch.empty((B, 2), dtype=torch.float64, device=x.device)
    if stats_out is not None:
        _require_cuda(stats_out)
        if stats_out.dtype != torch.float64 or stats_out.numel() < 2 * B * CT_LAB_STATS_STRIDE:
            raise CtHipError("stats_out must be float64 with >= 2*B*8 elements")
    ws = workspace(CT_WS_REINHARD_PERSIST, n, B, x.device)
    null = ctypes.c_void_p(0)
    if x.dtype == torch.uint8:
        fn = lib().ct_reinhard_psnr_u8
    else:
        fn = lib().ct_reinhard_persist_f32
    check(fn(_ptr(x), _ptr(r), _ptr(g) if g is not None else null, _ptr(out), _ptr(psnr_out) if g is not None else null, n, B,
                          _ptr(stats_out) if stats_out is not None else null, _ptr(ws), ws.numel(), _stream()))
    if verify:
        torch.cuda.synchronize()
        if int(ws[:4].view(torch.int32)[0].item()) != 0:
            raise CtHipError("persistent Reinhard launch: a workgroup of the grid never became resident (results are NaN)")
    o = out.view(tuple(target.shape))
    return (o, psnr_out) if g is not None else o


def mk(target, reference, decomposition="MK", out_dtype=torch.float64, out=None):
    """methods.linear.monge_kantorovitch_color_transfer on device tensors, B pairs per call, no host sync (ct_mk_*)."""
    x, _ = _as_batch(target)
    r, _ = _as_batch(reference)
    _require_cuda(x, r)
    if x.shape != r.shape or x.dtype != r.dtype:
        raise CtHipError("fused mk needs equal shapes/dtypes")
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    name = "ct_mk_%s_%s" % (_suffix(x), _suffix(out))
    if name not in SIGNATURES:
        raise CtHipError("no kernel for %s" % name)
    ws = workspace(CT_WS_REINHARD, n, B, x.device)
    mode = {"MK": 0, "sqrt": 1, "cholesky": 2}[decomposition]
    check(getattr(lib(), name)(_ptr(x), _ptr(r), _ptr(out), n, B, mode, _ptr(ws), ws.numel(), _stream()))
    return out.view(target.shape)


def frame_psnr(a, b):
    """Per-frame (mse, PSNR) of two float32 batches [B, ...] with data range 1 -> float64 [B, 2] (methods/__init__.py:32)."""
    _require_cuda(a, b)
    if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise CtHipError("frame_psnr needs two float32 tensors of one shape")
    B = a.shape[0]
    n = a.numel() // max(B, 1)
    out = torch.empty((B, 2), dtype=torch.float64, device=a.device)
    ws = workspace(CT_WS_LAB_STATS, n, B, a.device)
    check(lib().ct_frame_psnr_f32(_ptr(a), _ptr(b), n, B, _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


DISTORTIONS = {"identity": 0, "brightness": 1, "contrast": 2, "saturation": 3, "hue": 4, "gamma": 5}


def distort_u8(img, kind, param, want_u8=False):
    """torchvision.transforms.functional.adjust_<kind>(img, param) on a uint8 [3,H,W] device tensor (utils/data.py:12-22).
    Returns the distorted frame / 255 as float32 [3,H,W] (and the uint8 frame when want_u8)."""
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3 or img.shape[0] != 3 or not img.is_contiguous():
        raise CtHipError("distort_u8 needs a contiguous uint8 [3,H,W] device tensor")
    _check_device(img)
    h, w = img.shape[1], img.shape[2]
    out_f = torch.empty((3, h, w), dtype=torch.float32, device=img.device)
    out_u = torch.empty_like(img) if want_u8 else None
    ws = workspace(CT_WS_LAB_STATS, 0, 1, img.device, need=64)
    rc = lib().ct_distort_u8(_ptr(img), h, w, DISTORTIONS[kind] if isinstance(kind, str) else int(kind), float(param),
                             _ptr(out_u) if want_u8 else ctypes.c_void_p(0), _ptr(out_f), _ptr(ws), ws.numel(), _stream())
    if rc == -1:
        raise ValueError("distortion %r: parameter %r out of range" % (kind, param))      # torchvision raises ValueError too
    check(rc)
    return (out_f, out_u) if want_u8 else out_f


def regrain(img_in, img_col, nbits=(4, 16, 32, 64, 64, 64), out=None):
    """`_regrain(img_arr_in, img_arr_col, nbits)` of methods/iterative.py:62-117 on device tensors [H,W,3] (any float dtype;
    computed in float64).  Returns float64 [H,W,3]."""
    _require_cuda(img_in, img_col)
    if img_in.dim() != 3 or img_in.shape[2] != 3 or img_in.shape != img_col.shape:
        raise CtHipError("regrain needs two [H,W,3] tensors of one shape")
    if len(nbits) < 1 or len(nbits) > 8:
        raise CtHipError("regrain: nbits needs 1..8 entries")
    a, b = img_in.double().contiguous(), img_col.double().contiguous()
    h, w = a.shape[0], a.shape[1]
    if out is None:
        out = torch.empty_like(a)
    ws = workspace(CT_WS_LAB_STATS, 0, 1, a.device, need=lib().ct_regrain_workspace_bytes(h, w))
    nb = (ctypes.c_int * len(nbits))(*[int(v) for v in nbits])
    check(lib().ct_regrain_f64(_ptr(a), _ptr(b), _ptr(out), h, w, ctypes.cast(nb, ctypes.c_void_p), len(nbits), _ptr(ws), ws.numel(),
                               _stream()))
    return out


def _frame_metric(name, a, b):
    _require_cuda(a, b)
    if a.shape != b.shape or a.dim() != 4 or a.shape[1] != 3 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise CtHipError("%s needs two float32 [B,3,H,W] tensors of one shape" % name)
    B, _, h, w = a.shape
    out = torch.empty((B,), dtype=torch.float64, device=a.device)
    ws = workspace(CT_WS_LAB_STATS, 0, B, a.device, need=lib().ct_metric_workspace_bytes(h, w, B))
    check(getattr(lib(), name)(_ptr(a), _ptr(b), h, w, B, _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def frame_ssim(a, b):
    """Per-frame piq.ssim(a, b) (defaults, data range 1) of float32 [B,3,H,W] batches -> float64 [B] (methods/__init__.py:33)."""
    return _frame_metric("ct_frame_ssim_f32", a, b)


def frame_icid(a, b):
    """Per-frame utils.icid.icid(a, b) (perceptual intent) of float32 [B,3,H,W] batches -> float64 [B] (methods/__init__.py:35)."""
    return _frame_metric("ct_frame_icid_f32", a, b)


SIGNATURES.update({
    "ct_fft2d_c2c_f32": (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_fsim_pooled_size": (_c_int, [_c_int, _c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ct_fsim_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "ct_fsim_setup_f32": (_c_int, [_c_int, _c_int, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "ct_frame_fsim_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_sz, _c_p]),
})
_fsim_tables = {}                                  # (device, h, w) -> (filters [16, hp*wp] float32, consts [4, 3] float64)
CT_WS_FSIM = -7


def fft2d_(x, inverse=False):
    """In-place batched 2-D DFT of a complex64 tensor [..., H, W] (csrc/fft2d.hip; the transform inside frame_fsim): torch.fft.fft2,
    or -- inverse=True -- torch.fft.ifft2 without its 1 / (H W)."""
    if not x.is_cuda or x.dtype != torch.complex64 or not x.is_contiguous() or x.dim() < 2:
        raise CtHipError("fft2d_ needs a contiguous complex64 device tensor [..., H, W]")
    _check_device(x)
    h, w = x.shape[-2], x.shape[-1]
    planes = x.numel() // (h * w) if h * w else 0
    check(lib().ct_fft2d_c2c_f32(_ptr(x), h, w, planes, 1 if inverse else 0, _stream()))
    return x


def _fsim_ws(device, batch, h, w):
    need = lib().ct_fsim_workspace_bytes(batch, h, w)
    if need == 0:
        raise CtHipError("fsim: frames of %dx%d are too small (pooled size < 2)" % (h, w))
    ws = workspace(CT_WS_FSIM, 0, batch, device, need=need + 256)
    off = (-ws.data_ptr()) % 256
    return ws[off:off + need]


def frame_fsim(a, b):
    """Per-frame piq.fsim(a, b) (chromatic, piq defaults, data range 1) of float32 [B,3,H,W] batches -> float64 [B]
    (methods/__init__.py:34).  The filter bank of a frame size is built on the device at first use and cached."""
    _require_cuda(a, b)
    if a.shape != b.shape or a.dim() != 4 or a.shape[1] != 3 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise CtHipError("ct_frame_fsim_f32 needs two float32 [B,3,H,W] tensors of one shape")
    B, _, h, w = a.shape
    out = torch.empty((B,), dtype=torch.float64, device=a.device)
    if B == 0:
        return out
    ws = _fsim_ws(a.device, B, h, w)
    key = (str(a.device), h, w)
    with _lock:
        tab = _fsim_tables.get(key)
    if tab is None:
        hp, wp = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().ct_fsim_pooled_size(h, w, ctypes.byref(hp), ctypes.byref(wp)))
        filters = torch.empty((16, hp.value * wp.value), dtype=torch.float32, device=a.device)
        consts = torch.empty((4, 3), dtype=torch.float64, device=a.device)
        check(lib().ct_fsim_setup_f32(h, w, _ptr(filters), _ptr(consts), _ptr(ws), ws.numel(), _stream()))
        tab = (filters, consts)
        with _lock:
            _fsim_tables[key] = tab
    check(lib().ct_frame_fsim_f32(_ptr(a), _ptr(b), _ptr(out), B, h, w, _ptr(tab[0]), _ptr(tab[1]), _ptr(ws), ws.numel(), _stream()))
    return out


def mk_coef(stats_t, stats_r, decomposition="MK"):
    """On-device 3x3 algebra of MK (methods/linear.py:108-118): rgb_meancov records -> affine3x3 coefficient records."""
    mode = {"MK": 0, "sqrt": 1, "cholesky": 2}[decomposition]
    _require_cuda(stats_t, stats_r)
    b = stats_t.shape[0]
    coef = torch.empty((b, 16), dtype=torch.float64, device=stats_t.device)
    check(lib().ct_mk_coef_f64(_ptr(stats_t), _ptr(stats_r), mode, b, _ptr(coef), _stream()))
    return coef


def affine3x3(img, coef, out_dtype=torch.float64, out=None):
    """out = (x - mu_t) @ A + mu_r per image; coef float64 [B,16] = A[9], mu_t[3], mu_r[3], 0.
    Replaces methods/linear.py:80,122."""
    x, _ = _as_batch(img)
    _require_cuda(x, coef)
    B, n = x.shape[0], x.shape[1] * x.shape[2]
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    name = "ct_affine3x3_%s_%s" % (_suffix(x), _suffix(out))
    if name not in SIGNATURES:
        raise CtHipError("no kernel for %s" % name)
    check(getattr(lib(), name)(_ptr(x), _ptr(coef), _ptr(out), n, B, _stream()))
    return out.view(img.shape)


def idt(target, reference, rotations, bins=255, round_dr_f32=None, out=None, debug=False):
    """methods.iterative.iterative_distribution_transfer on device tensors (methods/iterative.py:8-59).

    target [H,W,3] or [B,H,W,3], reference likewise (its H x W may differ), float32 or float64 (same
    dtype); rotations: float64 numpy/tensor [n_iter,3,3] (shared by the batch) or [B,n_iter,3,3].
    Returns the float64 result (and a dict of device probe tensors when debug=True)."""
    import numpy as np
    x, _ = _as_batch(target)
    r, _ = _as_batch(reference)
    _require_cuda(x, r)
    if x.dtype != r.dtype or x.shape[0] != r.shape[0]:
        raise CtHipError("idt needs target/reference of one dtype and batch size")
    B, n_t, n_r = x.shape[0], x.shape[1] * x.shape[2], r.shape[1] * r.shape[2]
    rot = np.asarray(rotations.cpu().numpy() if isinstance(rotations, torch.Tensor) else rotations, dtype=np.float64)
    if rot.ndim == 3:
        rot = np.broadcast_to(rot, (B,) + rot.shape)
    n_iter = rot.shape[1]
    rinv = np.linalg.inv(rot)                       # host 3x3 inverses (the reference LU-solves, iterative.py:55)
    both = _upload_small(np.ascontiguousarray(np.stack([rot, rinv]).reshape(2, B, n_iter, 9)), x.device)
    if round_dr_f32 is None:
        round_dr_f32 = x.dtype == torch.float32
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    need = lib().ct_idt_workspace_bytes(B, n_iter, bins)
    if need == 0:
        raise CtHipError("idt: unsupported bins=%d / n_iter=%d" % (bins, n_iter))
    ws = workspace(CT_WS_IDT, 0, B, x.device, need=need)
    dbg_t, dbg_p = {}, ctypes.c_void_p(0)
    if debug:
        dbg_t = {"hist": torch.zeros((B, n_iter, 2, 3, bins), dtype=torch.int32, device=x.device),
                 "lut": torch.zeros((B, n_iter, 3, bins, 2), dtype=torch.float64, device=x.device),
                 "par": torch.zeros((B, n_iter, 3, 4), dtype=torch.float64, device=x.device),
                 "binidx": torch.zeros((B, n_iter, 3, n_t), dtype=torch.int16, device=x.device)}
        st = IdtDebug(dbg_t["hist"].data_ptr(), dbg_t["lut"].data_ptr(), dbg_t["par"].data_ptr(),
                      dbg_t["binidx"].data_ptr())
        dbg_p = ctypes.cast(ctypes.pointer(st), ctypes.c_void_p)
    fn = getattr(lib(), "ct_idt_" + _suffix(x))
    check(fn(_ptr(x), n_t, _ptr(r), n_r, B, _ptr(both[0]), _ptr(both[1]), n_iter, bins, int(bool(round_dr_f32)),
             _ptr(out), _ptr(ws), ws.numel(), dbg_p, _stream()))
    out = out.view(target.shape)
    return (out, dbg_t) if debug else out


# ------------------------------------------------------------------------------------------------
# DCMCS3DI building blocks (csrc/cnn.hip)
# ------------------------------------------------------------------------------------------------
_c_ll = ctypes.c_longlong
SIGNATURES.update({
    "ct_conv2d_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll,
                               _c_ll, _c_int, _c_int, _c_p]),
    "ct_pam_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "ct_conv2d_split_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                     _c_int, _c_int, _c_ll, _c_ll, _c_ll, _c_ll, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p,
                                     _c_p, _c_p, _c_ll, _c_p]),
    "ct_conv_split_scratch_bytes": (_c_sz, []),
    "ct_pam_attend_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_pam_valid_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_sz, _c_p]),
})


# ---- convolution arithmetic --------------------------------------------------------------------------------------
# "split": float32 operands as three bf16 pieces, six bf16 MFMAs per product (csrc/conv_split.hip; float32-grade
#          accuracy, 2.67x the matrix rate).  "exact": v_mfma_f32_32x32x2_f32, bitwise an fmaf chain (csrc/cnn.hip).
# Geometries the split kernel does not cover (stride 2, 7x7, W % 4 != 0, unaligned views) always run "exact".
_conv_mode = os.environ.get("CT_HIP_CONV", "split")
# split mode: two fp16 pieces / three MFMAs per product (power-of-two scales per layer and per staged tile / row) instead of three
# bf16 pieces / six MFMAs, in the weight-stationary kernel (csrc/conv_ws.hip) and the tile kernel (csrc/conv_split.hip);
# CT_HIP_CONV_WS16=0 switches back to the bf16 form
_ws16 = os.environ.get("CT_HIP_CONV_WS16", "1") != "0"


def set_conv_ws16(on):
    global _ws16
    _ws16 = bool(on)


def conv_ws16():
    return _ws16


def set_conv_mode(mode):
    global _conv_mode
    if mode not in ("split", "exact"):
        raise ValueError("conv mode must be 'split' or 'exact'")
    _conv_mode = mode


def conv_mode():
    return _conv_mode


def pack_conv_weight_split(weight, bias):
    """Conv2d parameters -> ct_conv2d_split_f32 operands: bf16 bit patterns (int16)
    [ceil(cout/64)][ceil(cin/16)][kh*kw][piece hi/mid/lo][m 0..1][k-half 0..1][cout%32][8 channels], and the bias
    zero-padded to 64*ceil(cout/64)."""
    cout, cin, kh, kw = weight.shape
    g, nc = (cout + 63) // 64, (cin + 15) // 16
    w = torch.zeros((g * 64, nc * 16, kh, kw), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.detach().float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    r1 = torch.where(torch.isfinite(r1), r1, torch.zeros_like(r1))
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    pieces = torch.stack([hi, mid, lo], dim=0).view(torch.int16)           # [3][coutp][cinp][kh][kw]
    pieces = pieces.reshape(3, g, 2, 32, nc, 2, 8, kh * kw)                # piece, g, m, r, chunk, h, j, tap
    ws = pieces.permute(1, 4, 7, 0, 2, 5, 3, 6).contiguous()               # g, chunk, tap, piece, m, h, r, j
    b = torch.zeros(g * 64, dtype=torch.float32, device=weight.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return ws, b


def pack_conv_weight_split16(weight):
    """3x3 Conv2d weight (32 < cin <= 64) -> ct_conv3x3_ws16_f32 operand: fp16 bit patterns (int16)
    [ceil(cout/64)][ceil(cin/16)][9][piece hi/lo][m 0..1][k-half 0..1][cout%32][8 channels] of weight * 2^w_exp, and w_exp
    (the largest |weight| lands in [2^11, 2^12): both pieces of all but the tiniest weights are normal fp16 numbers)."""
    cout, cin, kh, kw = weight.shape
    g, nc = (cout + 63) // 64, (cin + 15) // 16
    w = torch.zeros((g * 64, nc * 16, kh, kw), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.detach().float()
    amax = float(w.abs().max())
    w_exp = 0 if not (amax > 0 and amax < float("inf")) else 12 - (int(np.floor(np.log2(amax))) + 1)
    w_exp = max(-100, min(100, w_exp))
    ws = w * (2.0 ** w_exp)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    pieces = torch.stack([hi, lo], dim=0).view(torch.int16)                # [2][coutp][cinp][kh][kw]
    pieces = pieces.reshape(2, g, 2, 32, nc, 2, 8, kh * kw)                # piece, g, m, r, chunk, h, j, tap
    return pieces.permute(1, 4, 7, 0, 2, 5, 3, 6).contiguous(), w_exp      # g, chunk, tap, piece, m, h, r, j


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def pack_conv_weight_wino16(weight):
    """3x3 Conv2d weight (32 < cin <= 64) -> ct_conv3x3_wino16_f32 operand: the Winograd F(2x2, 3x3) image U = G g G^T (float64,
    rounded once to float32), times 2^w_exp (largest |U| in [2^11, 2^12)), as fp16 hi / lo bit patterns (int16)
    [ceil(cout/64)][16 positions][4 cout blocks][2 cin chunks][piece][64 lanes][8]: lane l of a fragment holds cout 16 mb + l % 16,
    cin 32 kc + 8 (l / 16) + 0..7 (the A operand of v_mfma_f32_16x16x32_f16).  Returns (image, w_exp)."""
    cout, cin, kh, kw = weight.shape
    assert (kh, kw) == (3, 3) and cin <= 64
    g = (cout + 63) // 64
    w = torch.zeros((g * 64, 64, 3, 3), dtype=torch.float64, device=weight.device)
    w[:cout, :cin] = weight.detach().double()
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=weight.device)
    u = torch.einsum("ij,kcjl,ml->kcim", G, w, G).float()                 # [coutp][64][4][4]
    amax = float(u.abs().max())
    w_exp = 0 if not (amax > 0 and amax < float("inf")) else 12 - (int(np.floor(np.log2(amax))) + 1)
    w_exp = max(-100, min(100, w_exp))
    us = u * (2.0 ** w_exp)
    hi = us.to(torch.float16)
    lo = (us - hi.float()).to(torch.float16)
    pieces = torch.stack([hi, lo], dim=0).view(torch.int16)                # [piece][coutp][64][4][4]
    pieces = pieces.reshape(2, g, 4, 16, 2, 4, 8, 16)                      # piece, g, mb, m, kc, kblk, e, p
    img = pieces.permute(1, 7, 2, 4, 0, 5, 3, 6).contiguous()              # g, p, mb, kc, piece, kblk, m, e  (lane = 16 kblk + m)
    return img.reshape(g, 16, 4, 2, 2, 64, 8), w_exp


_wino = os.environ.get("CT_HIP_CONV_WINO", "1") not in ("0", "")


def set_conv_wino(on):
    """True (default; env CT_HIP_CONV_WINO=0 turns it off): the 3x3 convolutions that ct_conv3x3_ws16_f32 would take run as
    Winograd F(2x2, 3x3) (ct_conv3x3_wino16_f32: 2.25x fewer matrix instructions, float32-grade); False: the direct kernel."""
    global _wino
    _wino = bool(on)


def conv_wino():
    return _wino


def set_conv_wino_form(form):
    """Which Winograd kernel ct_conv3x3_wino16_f32 launches (include/ct_hip.h: ct_set_conv_wino_form): 0 = the four-wave pipelined
    kernel of round 6 (csrc/conv_wino4.hip, default), 1 = the eight-wave kernel of round 5 (csrc/conv_wino.hip)."""
    check(lib().ct_set_conv_wino_form(int(form)))


def _ws16_ok(x, split, kh, kw):
    return _ws16 and (kh, kw) == (3, 3) and 32 < x.shape[1] <= 64 and len(split) > 2 and split[2] is not None


def _split_operands(weight, bias):
    """(bf16 pieces, padded bias, fp16 image or None): what travels with a packed convolution weight as `_ct_split`"""
    cout, cin, kh, kw = weight.shape
    w16 = pack_conv_weight_split16(weight) if (kh, kw) in ((3, 3), (1, 1), (1, 5), (5, 1), (2, 2)) else None
    wq = pack_conv_weight_wino16(weight) if (kh, kw) == (3, 3) and 32 < cin <= 64 else None      # Winograd image (ct_conv3x3_wino16_f32)
    return pack_conv_weight_split(weight, bias) + (w16, wq)


def _split_ok(x, out, residual, kh, kw, stride, ph, pw):
    if _conv_mode != "split" or stride != 1 or (kh, kw) not in ((3, 3), (1, 1), (1, 5), (5, 1)) or (ph, pw) != (kh // 2, kw // 2):
        return False
    if x.shape[3] % 4:
        return False
    for t in (x, out, residual):
        if t is not None and (t.data_ptr() % 16 or t.stride(0) % 4):
            return False
    return True


_sk_cache = {}                                   # (device index, stream) -> zero-initialised stream-K scratch of ct_conv2d_split_f32
_stream_k = True


def set_conv_stream_k(on):
    """False: the tile convolution gets no scratch, i.e. every workgroup computes whole (tile, 64-channel) units (include/ct_hip.h:
    ct_conv2d_split_f32, scratch == NULL); True (default): badly quantised launches share units between neighbouring workgroups."""
    global _stream_k
    _stream_k = bool(on)


def conv_stream_k_state(device=None):
    """(nonzero flag words, consumers that gave up) of the current stream's stream-K scratch -- both 0 between launches; None
    before the first launch on this stream."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    buf = _sk_cache.get((device.index, torch.cuda.current_stream(device).cuda_stream))
    if buf is None:
        return None
    words = buf[:4096].view(torch.int32)
    return int(words[:1000].ne(0).sum()), int(words[1000])


def _conv_scratch(device):
    """The stream-K scratch of the tile convolution for the current stream (include/ct_hip.h: all zero before its first use, then
    owned by the launches of one stream, which leave its flag words zero again).  None while the stream is being captured and
    no scratch exists for it yet: an allocation made inside a capture belongs to that graph's private pool (its zero fill is a
    node of that graph only), so a later graph on the same capture stream would share memory the allocator may already have
    handed out again -- such launches run without stream-K instead (every workgroup computes whole units, same results)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    need = lib().ct_conv_split_scratch_bytes()      # outside the lock: lib() takes it on first use
    with _lock:
        buf = _sk_cache.get(key)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            buf = torch.zeros(need, dtype=torch.uint8, device=device)
            _sk_cache[key] = buf
    return buf


def conv_scratch_prepare(device=None, stream=None):
    """Create the stream-K scratch of `stream` (default: the current one) OUTSIDE any capture, e.g. for the stream a
    torch.cuda.graph() block is about to capture on, so that the captured convolutions keep stream-K."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    if stream is None:
        return _conv_scratch(device)
    with torch.cuda.stream(stream):
        return _conv_scratch(device)


def _conv_split(x, split, cout, kh, kw, act, residual, clamp, out, x2=None, x3=None, res_pre=False, post=None):
    ws, b64 = split[0], split[1]
    f16, w_exp = 0, 0
    if _ws16 and len(split) > 2 and split[2] is not None:      # two fp16 pieces (default): the fp16 image replaces the bf16 one
        (ws, w_exp), f16 = split[2], 1
    n, cin1, h, w = x.shape
    cin2 = cin1 + (x2.shape[1] if x2 is not None else 0)
    cin = cin2 + (x3.shape[1] if x3 is not None else 0)
    rs = _nchw_bstride(residual) if residual is not None else 0
    post_op, p1, p2 = (0, None, None) if post is None else post
    if post_op and not f16:
        raise CtHipError("a fused post-op needs the fp16 form of the split kernel")
    if x2 is None and not res_pre and not post_op and _ws16_ok(x, split, kh, kw):
        if _wino and len(split) > 3 and split[3] is not None and h * w * 256 < (1 << 32):      # 32-bit byte offsets over 64 output planes; larger: the direct kernel
            wq, wq_exp = split[3]
            check(lib().ct_conv3x3_wino16_f32(_ptr(x), _ptr(wq), int(wq_exp), _ptr(b64), _opt(residual), _ptr(out), n, cin, cout, h, w,
                                              _nchw_bstride(x), _nchw_bstride(out), rs, int(act), int(bool(clamp)), _stream()))
            return out
        w16, w_exp = split[2]
        check(lib().ct_conv3x3_ws16_f32(_ptr(x), _ptr(w16), int(w_exp), _ptr(b64), _opt(residual), _ptr(out), n, cin, cout, h, w,
                                        _nchw_bstride(x), _nchw_bstride(out), rs, int(act), int(bool(clamp)), _stream()))
        return out
    scratch = _conv_scratch(x.device) if (f16 and _stream_k) else None
    check(lib().ct_conv2d_split_f32(_ptr(x), _opt(x2), cin1, _opt(x3), cin2, _ptr(ws), _ptr(b64), _opt(residual), _ptr(out), n, cin,
                                    cout, h, w, kh, kw, _nchw_bstride(x), _nchw_bstride(x2) if x2 is not None else 0,
                                    _nchw_bstride(x3) if x3 is not None else 0, _nchw_bstride(out), rs, int(act), int(bool(clamp)),
                                    int(bool(res_pre)), f16, int(w_exp), int(post_op), _opt(p1), _opt(p2), _opt(scratch),
                                    scratch.numel() if scratch is not None else 0, _stream()))
    return out


def pack_conv_weight(weight, bias):
    """torch Conv2d parameters -> the MFMA A-operand layout of ct_conv2d_f32 (include/ct_hip.h):
    wp[tap][cin_pair][2][32*ceil(cout/32)], bias zero padded."""
    cout, cin, kh, kw = weight.shape
    assert kh == kw and kh in (1, 3)
    coutp = 32 * ((cout + 31) // 32)
    cinp = 2 * ((cin + 1) // 2)
    w = torch.zeros((coutp, cinp, kh, kw), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.detach().float()
    wp = w.permute(2, 3, 1, 0).reshape(kh * kw, cinp // 2, 2, coutp).contiguous()
    b = torch.zeros(coutp, dtype=torch.float32, device=weight.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    wp._ct_split = _split_operands(weight, bias)      # operands of the split kernels travel with the packing
    return wp, b


def _nchw_bstride(t):
    n, c, h, w = t.shape
    if t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w:
        raise CtHipError("conv2d needs NCHW tensors with dense planes (channel slices are fine)")
    return t.stride(0)


def conv2d(x, wp, bias, cout, ksize, act=0, residual=None, clamp=False, out=None, x2=None, x3=None):
    """Conv2d(ksize, padding=ksize//2) + bias [+ LeakyReLU(0.01)] [+ residual] [clamp 0..1], float32 NCHW.
    x2 / x3: further input tensors whose channels follow x's -- torch.cat([x, x2, x3], 1) without the copy when the split kernel
    takes the convolution (channel counts of x and x + x2 multiples of 16); otherwise the concatenation is materialised here."""
    if x2 is not None:
        split = getattr(wp, "_ct_split", None)
        c1, c2 = x.shape[1], x.shape[1] + x2.shape[1]
        ok = (split is not None and c1 % 16 == 0 and (x3 is None or c2 % 16 == 0) and
              all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and
                                t[0].is_contiguous()) for t in (x2, x3)))
        if out is None:
            out = torch.empty((x.shape[0], cout, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        if ok and _split_ok(x, out, residual, ksize, ksize, 1, ksize // 2, ksize // 2):
            return _conv_split(x, split, cout, ksize, ksize, act, residual, clamp, out, x2=x2, x3=x3)
        x = torch.cat([t for t in (x, x2, x3) if t is not None], dim=1)
    if x.is_cuda:
        _check_device(x)
    if not x.is_cuda or x.dtype != torch.float32:
        raise CtHipError("conv2d needs float32 CUDA tensors (no CPU path)")
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    split = getattr(wp, "_ct_split", None)
    if split is not None and _split_ok(x, out, residual, ksize, ksize, 1, ksize // 2, ksize // 2):
        return _conv_split(x, split, cout, ksize, ksize, act, residual, clamp, out)
    rs = _nchw_bstride(residual) if residual is not None else 0
    check(lib().ct_conv2d_f32(_ptr(x), _ptr(wp), _ptr(bias), _ptr(residual) if residual is not None else _c_p(0),
                              _ptr(out), n, cin, cout, h, w, ksize, _nchw_bstride(x), _nchw_bstride(out), rs, int(act),
                              int(bool(clamp)), _stream()))
    return out


def conv2d_rows(x, wp, bias, cout, ksize, act=0, out=None, c0=0, channels=None, raw=None):
    """conv2d whose result is written as token rows: out[n*H + y, x, c0 + co] of a [N*H, W, channels] tensor (the layout the
    streaming attention reads), so the NCHW tensor and its transpose are never made.  Returns None when the split kernel cannot
    take the convolution (exact mode, W % 4): the caller then runs conv2d + the transpose.
    raw = (Conv2d weight, bias or None): with it a 1x1 convolution of 64 input channels takes the streaming float32 kernel
    ct_conv1x1_rows_f32 (csrc/conv1x1_rows.hip) in every convolution mode."""
    split = getattr(wp, "_ct_split", None)
    n, cin, h, w = x.shape
    if (raw is not None and ksize == 1 and cin == 64 and cout <= 64 and cout % 4 == 0 and x.is_cuda and x.dtype == torch.float32 and
            x.stride(3) == 1 and x.stride(2) == w and x.stride(1) == h * w and act <= 4):
        _check_device(x)
        channels = int(channels if channels is not None else (out.shape[2] if out is not None else cout))
        if out is None:
            out = torch.empty((n * h, w, channels), dtype=torch.float32, device=x.device)
        if (out.shape != (n * h, w, channels) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device or
                channels % 4 or c0 % 4 or c0 + cout > channels):
            raise CtHipError("conv2d_rows: out must be a contiguous float32 [N*H, W, channels] tensor, channels / c0 / cout multiples of 4")
        wt = raw[0].detach().reshape(cout, 64).contiguous().float()
        bs = raw[1].detach().contiguous().float() if raw[1] is not None else torch.zeros(cout, dtype=torch.float32, device=x.device)
        check(lib().ct_conv1x1_rows_f32(_ptr(x), _ptr(wt), _ptr(bs), _ptr(out), n, cin, cout, h, w, _nchw_bstride(x), channels, int(c0),
                                        int(act), _stream()))
        return out
    if split is None or not x.is_cuda or x.dtype != torch.float32 or not _split_ok(x, None, None, ksize, ksize, 1, ksize // 2, ksize // 2):
        return None
    _check_device(x)
    channels = int(channels if channels is not None else (out.shape[2] if out is not None else cout))
    if out is None:
        out = torch.empty((n * h, w, channels), dtype=torch.float32, device=x.device)
    if (out.shape != (n * h, w, channels) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device or
            channels % 4 or c0 % 4 or cout % 4 or c0 + cout > channels):
        raise CtHipError("conv2d_rows: out must be a contiguous float32 [N*H, W, channels] tensor, channels / c0 / cout multiples of 4")
    ws, f16, w_exp = split[0], 0, 0
    if _ws16 and len(split) > 2 and split[2] is not None:
        (ws, w_exp), f16 = split[2], 1
    check(lib().ct_conv2d_split_rows_f32(_ptr(x), _ptr(ws), _ptr(split[1]), _ptr(out), n, cin, cout, h, w, ksize, ksize,
                                         _nchw_bstride(x), channels, int(c0), int(act), f16, int(w_exp), _stream()))
    return out


def pam_attend(q, k, v, rgb, want_att=False):
    """softmax(q.k/c) @ [v | rgb] per image row (pasmnet/attention.py:39-41, utils.py:30,123-125)."""
    for t in (q, k, v, rgb):
        if t.is_cuda:
            _check_device(t)
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise CtHipError("pam_attend needs contiguous float32 CUDA tensors")
    n, c, h, w = q.shape
    cv = v.shape[1]
    out_v = torch.empty_like(v)
    out_rgb = torch.empty_like(rgb)
    att = torch.empty((n, h, w, w), dtype=torch.float32, device=q.device) if want_att else None
    check(lib().ct_pam_attend_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(rgb), _ptr(out_v), _ptr(out_rgb),
                                  _ptr(att) if att is not None else _c_p(0), n, c, cv, h, w, _stream()))
    return out_v, out_rgb, att


def pam_valid(q, k, want_att=False):
    """valid mask (as 0/1 float [n,1,h,w]) + pre-threshold column sums of softmax(q.k/c) (utils.py:31,34-35)."""
    for t in (q, k):
        if t.is_cuda:
            _check_device(t)
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise CtHipError("pam_valid needs contiguous float32 CUDA tensors")
    n, c, h, w = q.shape
    valid = torch.empty((n, 1, h, w), dtype=torch.float32, device=q.device)
    colsum = torch.empty((n, 1, h, w), dtype=torch.float32, device=q.device)
    att = torch.empty((n, h, w, w), dtype=torch.float32, device=q.device) if want_att else None
    need = lib().ct_pam_workspace_bytes(n, h, w)
    ws = workspace(-1, 0, 0, q.device, need=need)
    check(lib().ct_pam_valid_f32(_ptr(q), _ptr(k), _ptr(valid), _ptr(colsum), _ptr(att) if att is not None else _c_p(0),
                                 n, c, h, w, _ptr(ws), ws.numel(), _stream()))
    return valid, colsum, att


# ------------------------------------------------------------------------------------------------
# GMFlow building blocks (csrc/gmflow.hip)
# ------------------------------------------------------------------------------------------------
_c_f = ctypes.c_float
SIGNATURES.update({
    "ct_gconv2d_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 10 + [_c_ll, _c_ll, _c_int, _c_p]),
    "ct_conv2d_split_rows_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 7 + [_c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_conv3x3_ws16_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_ll, _c_ll, _c_ll, _c_int, _c_int, _c_p]),
    "ct_conv3x3_wino16_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_ll, _c_ll, _c_ll, _c_int, _c_int, _c_p]),
    "ct_set_conv_wino_form": (_c_int, [_c_int]),
    "ct_conv1x1_rows_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_ll, _c_int, _c_int, _c_int, _c_p]),
    "ct_instance_norm_workspace_bytes": (ctypes.c_size_t, [_c_int]),
    "ct_instance_norm_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_f, _c_int, _c_p, ctypes.c_size_t, _c_p]),
    "ct_eltwise_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_f, _c_p]),
    "ct_space_to_depth2_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_p]),
    "ct_linear_tokens_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_p]),
    "ct_linear_tokens_split_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_p]),
    "ct_layernorm128_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_p]),
    "ct_linear_ws16_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p]),
    "ct_attention_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "ct_attention_tokens_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_int, _c_p,
                                         ctypes.c_size_t, ctypes.c_longlong, _c_p]),
    "ct_nchw_to_rows_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_int, _c_int, _c_p]),
    "ct_rows_to_nchw_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_int, _c_int, _c_p]),
    "ct_attention_rows64_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p]),
    "ct_attention_colsum64_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p]),
    "ct_local_corr_softmax_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_local_corr_flow_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_local_attn_prop_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_bilinear_resize_f32": (_c_int, [_c_p, _c_p] + [_c_int] * 6 + [_c_f, _c_f, _c_p]),
    "ct_flow_warp_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_convex_upsample_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "ct_fb_check_f32": (_c_int, [_c_p] * 6 + [_c_int, _c_int, _c_int, _c_f, _c_f, _c_p]),
})

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_GELU = 0, 1, 2, 3, 4, 6


def _f32c(*ts):
    for t in ts:
        if t is None:
            continue
        if t.is_cuda:
            _check_device(t)
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise CtHipError("needs contiguous float32 CUDA tensors (no CPU path)")


def _opt(t):
    return _ptr(t) if t is not None else _c_p(0)


def pack_gconv_weight(weight, bias):
    """Conv2d parameters -> ct_gconv2d_f32 layout: output channels in groups of 64 (zero padded),
    wp[ceil(cout/64)][kh*kw][ceil(cin/2)][2][64]; bias zero padded to 64*ceil(cout/64) (zeros when the conv has none)."""
    cout, cin, kh, kw = weight.shape
    coutp, cinp = 64 * ((cout + 63) // 64), 2 * ((cin + 1) // 2)
    w = torch.zeros((coutp, cinp, kh, kw), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.detach().float()
    # [g][co64][cin_pair][2][kh][kw] -> [g][kh][kw][cin_pair][2][co64]
    wp = w.reshape(coutp // 64, 64, cinp // 2, 2, kh, kw).permute(0, 4, 5, 2, 3, 1).contiguous()
    wp = wp.reshape(coutp // 64, kh * kw, cinp // 2, 2, 64)
    b = torch.zeros(coutp, dtype=torch.float32, device=weight.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    if kh * kw <= 9:
        wp._ct_split = _split_operands(weight, bias)
    if (kh, kw) == (3, 3):
        wp._ct_src = (weight, bias)               # a stride-2 use builds its space-to-depth form from these (first use, cached)
    return wp, b


def space_to_depth2(x):
    """[n, c, h, w] -> [n, 4c, h/2, w/2], channel (2 sy + sx) c + ch = x[:, ch, sy::2, sx::2] (ct_space_to_depth2_f32).
    No cache here (round 4 kept the last image keyed on the tensor's identity and `_version`: inference tensors have no version
    counter, and writers that go through raw pointers -- this library's own out= entries -- do not bump it): a caller whose two
    stride-2 convolutions read one input makes the image once and hands it to both (gconv2d(..., s2d=...))."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or not x[0].is_contiguous():
        raise CtHipError("space_to_depth2 needs a float32 CUDA tensor [n, c, h, w] with dense images (no CPU path)")
    _check_device(x)
    n, c, h, w = x.shape
    out = torch.empty((n, 4 * c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(lib().ct_space_to_depth2_f32(_ptr(x), _ptr(out), n, c, h, w, _nchw_bstride(x), _stream()))
    return out


def s2d_ok(x):
    """True when a stride-2 3x3 'same' / 1x1 convolution of x can take the tile kernel over space_to_depth2(x) (fp16 form)"""
    return (_conv_mode == "split" and _ws16 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] % 2 == 0 and
            x.shape[3] % 8 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0 and x[0].is_contiguous())


def _split_s2d(wp):
    """the 2x2 / 4c form of a 3x3 stride-2 convolution's weight (include/ct_hip.h: ct_space_to_depth2_f32), packed like _ct_split"""
    hit = getattr(wp, "_ct_split_s2d", None)
    weight, bias = wp._ct_src
    ver = (weight._version, weight.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()))
    if hit is None or hit[0] != ver:
        cout, cin = weight.shape[:2]
        w2 = torch.zeros((cout, 4, cin, 2, 2), dtype=torch.float32, device=weight.device)
        wd = weight.detach().float()
        taps = {0: (0, 1), 1: (1, 0), 2: (1, 1)}           # k -> (block offset index, sub-position): 2 o + k - 1 = 2 (o + b - 1) + s
        for ky, (by, sy) in taps.items():
            for kx, (bx, sx) in taps.items():
                w2[:, 2 * sy + sx, :, by, bx] = wd[:, :, ky, kx]
        hit = (ver, _split_operands(w2.reshape(cout, 4 * cin, 2, 2), bias))
        wp._ct_split_s2d = hit
    return hit[1]


def gconv2d(x, wp, bias, cout, ksize, stride=1, padding=0, act=ACT_NONE, out=None, x2=None, residual=None, addend=None, post=None, s2d=None):
    """s2d: space_to_depth2(x) made by the caller (stride-2 convolutions that share their input); x2: optional second input tensor whose channels follow x's (torch.cat([x, x2], 1) without the copy when the
    split-bf16 kernel takes the convolution; otherwise the concatenation is materialised here).  residual: added to the
    result (act must be ACT_NONE: MBConvBlock's identity skip).  addend: a tensor of the output's shape added BEFORE the
    activation (a pre-computed part of the convolution); split kernel only -- CtHipError otherwise.  post (with addend, fp16 form):
    (1, p1, None) = the activated result times p1; (2, z, h) = (1 - z) * h + z * result -- the GRU's two elementwise steps."""
    kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    if residual is not None and (act != ACT_NONE or x2 is not None or addend is not None):
        raise CtHipError("gconv2d: a residual needs act=ACT_NONE, a single input and no addend")
    if addend is not None:
        split = getattr(wp, "_ct_split", None)
        n, c1, h, w = x.shape
        if out is None:
            out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        ok = (split is not None and bias is not None and addend.shape == out.shape and addend.dtype == torch.float32 and
              _split_ok(x, out, addend, kh, kw, stride, ph, pw) and
              (x2 is None or (c1 % 16 == 0 and x2.data_ptr() % 16 == 0 and x2.stride(0) % 4 == 0)))
        if not ok:
            raise CtHipError("gconv2d: an addend needs the split kernel (conv mode 'split', stride 1, 'same' padding, W % 4 == 0)")
        if post is not None:
            if not _ws16 or any(t is not None and (t.shape != out.shape or not t.is_contiguous() or t.dtype != torch.float32) for t in post[1:]):
                raise CtHipError("gconv2d: post-op operands must be contiguous float32 tensors of the output's shape (fp16 form only)")
            if not out.is_contiguous():
                raise CtHipError("gconv2d: a post-op needs a contiguous output")
        return _conv_split(x, split, cout, kh, kw, act, addend, False, out, x2=x2, res_pre=True, post=post)
    if x2 is not None:
        split = getattr(wp, "_ct_split", None)
        n, c1, h, w = x.shape
        if (split is not None and bias is not None and c1 % 16 == 0 and x2.data_ptr() % 16 == 0 and x2.stride(0) % 4 == 0 and
                _split_ok(x, out if out is not None else x, None, kh, kw, stride, ph, pw)):
            if out is None:
                out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
            return _conv_split(x, split, cout, kh, kw, act, None, False, out, x2=x2)
        x = torch.cat([x, x2], dim=1)
    n, cin, h, w = x.shape
    ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    if out is None:
        out = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
    split = getattr(wp, "_ct_split", None)
    # stride 2, 3x3 "same" or 1x1: the MFMA tile kernel over the space-to-depth image of x (fp16 form)
    if (stride == 2 and split is not None and bias is not None and residual is None and cout > 4 and s2d_ok(x) and
            out.data_ptr() % 16 == 0 and out.stride(0) % 4 == 0 and
            (((kh, kw, ph, pw) == (3, 3, 1, 1) and hasattr(wp, "_ct_src")) or (kh, kw, ph, pw) == (1, 1, 0, 0))):
        if s2d is None:
            s2d = space_to_depth2(x)
        elif s2d.shape != (n, 4 * cin, h // 2, w // 2) or s2d.dtype != torch.float32 or not s2d.is_contiguous():
            raise CtHipError("gconv2d: s2d must be space_to_depth2(x)")
        if kh == 1:
            return _conv_split(s2d[:, :cin], split, cout, 1, 1, act, None, False, out)
        return _conv_split(s2d, _split_s2d(wp), cout, 2, 2, act, None, False, out)
    # cout <= 4 (the flow head's 256 -> 2): ct_gconv2d_f32's direct kernel instead of a 64-output-channel tile
    if split is not None and bias is not None and (cout > 4 or residual is not None) and _split_ok(x, out, residual, kh, kw, stride, ph, pw):
        return _conv_split(x, split, cout, kh, kw, act, residual, False, out)
    check(lib().ct_gconv2d_f32(_ptr(x), _ptr(wp), _opt(bias), _ptr(out), n, cin, cout, h, w, kh, kw, stride, ph, pw,
                               _nchw_bstride(x), _nchw_bstride(out), int(act), _stream()))
    if residual is not None:
        return eltwise(0, out, residual)
    return out


def instance_norm(x, mode=0, skip=None, eps=1e-5):
    _f32c(x, skip)
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    need = lib().ct_instance_norm_workspace_bytes(n * c)
    ws = workspace(-3, 0, 0, x.device, need=need)
    check(lib().ct_instance_norm_f32(_ptr(x), _opt(skip), _ptr(y), n * c, h * w, eps, mode, _ptr(ws), need, _stream()))
    return y


def eltwise(op, a, b=None, c=None, plane=1, chans=1, split=0, s0=1.0):
    _f32c(a, b, c)
    y = torch.empty_like(a)
    check(lib().ct_eltwise_f32(_ptr(a), _opt(b), _opt(c), _ptr(y), a.numel(), op, plane, chans, split, float(s0), _stream()))
    return y


# ------------------------------------------------------------------------------------------------
# f4: layers of DMSCT's EfficientNet-B2 / U-Net (csrc/unet.hip)
# ------------------------------------------------------------------------------------------------
ACT_SWISH = 5
SIGNATURES.update({
    "ct_gconv2d_pad_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 12 + [_c_ll, _c_ll, _c_int, _c_p]),
    "ct_dwconv_tiles": (_c_int, [_c_int, _c_int]),
    "ct_dwconv_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 11 + [_c_p, _c_p]),
    "ct_se_gate_f32": (_c_int, [_c_p, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "ct_scale_planes_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_p]),
    "ct_upsample2_concat_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
})


def gconv2d_pad(x, wp, bias, cout, ksize, stride, pad_top_left, out_size, act=ACT_NONE):
    """generic convolution with explicit (top, left) zero padding and output size (TF-"SAME" static padding)"""
    _f32c(x)
    n, cin, h, w = x.shape
    out = torch.empty((n, cout, out_size[0], out_size[1]), dtype=torch.float32, device=x.device)
    check(lib().ct_gconv2d_pad_f32(_ptr(x), _ptr(wp), _opt(bias), _ptr(out), n, cin, cout, h, w, ksize, ksize, stride, pad_top_left[0],
                                   pad_top_left[1], out_size[0], out_size[1], _nchw_bstride(x), _nchw_bstride(out), int(act), _stream()))
    return out


def dwconv(x, weight, bias, ksize, stride, pad_top_left, out_size, act=ACT_SWISH, want_sums=False):
    """depthwise convolution, weight [C, k*k] and bias [C] with the BatchNorm folded in; -> out (, tile sums [N, C, tiles])"""
    _f32c(x, weight, bias)
    n, c, h, w = x.shape
    out = torch.empty((n, c, out_size[0], out_size[1]), dtype=torch.float32, device=x.device)
    sums = None
    if want_sums:
        sums = torch.empty((n, c, lib().ct_dwconv_tiles(out_size[0], out_size[1])), dtype=torch.float32, device=x.device)
    check(lib().ct_dwconv_f32(_ptr(x), _ptr(weight), _ptr(bias), _ptr(out), n, c, h, w, ksize, stride, pad_top_left[0], pad_top_left[1],
                              out_size[0], out_size[1], int(act), _opt(sums), _stream()))
    return (out, sums) if want_sums else out


def se_gate(tile_sums, plane, w_reduce, b_reduce, w_expand, b_expand):
    """squeeze-and-excitation gate [N, C] from the tile sums of the depthwise output"""
    _f32c(tile_sums, w_reduce, b_reduce, w_expand, b_expand)
    n, c, tiles = tile_sums.shape
    gate = torch.empty((n, c), dtype=torch.float32, device=tile_sums.device)
    check(lib().ct_se_gate_f32(_ptr(tile_sums), tiles, plane, _ptr(w_reduce), _ptr(b_reduce), _ptr(w_expand), _ptr(b_expand), _ptr(gate),
                               n, c, w_reduce.shape[0], _stream()))
    return gate


def scale_planes_(x, gate):
    """x[n, c] *= gate[n, c] in place"""
    _f32c(x, gate)
    n, c, h, w = x.shape
    check(lib().ct_scale_planes_f32(_ptr(x), _ptr(gate), n * c, h * w, _stream()))
    return x


def upsample2_concat(x, skip=None):
    """cat([nearest-x2(x), skip], dim=1)"""
    _f32c(x, skip)
    n, cx, h, w = x.shape
    cs = 0 if skip is None else skip.shape[1]
    if skip is not None and tuple(skip.shape) != (n, cs, 2 * h, 2 * w):
        raise CtHipError("upsample2_concat: skip must be [N, Cs, 2H, 2W]")
    out = torch.empty((n, cx + cs, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    check(lib().ct_upsample2_concat_f32(_ptr(x), _opt(skip), _ptr(out), n, cx, cs, h, w, _stream()))
    return out


def pack_linear_weight_split(weight):
    """nn.Linear weight [N, K] (K % 32 == 0) -> ct_linear_tokens_split_f32's operand: bf16 bit patterns (int16)
    [ceil(N/128)][K/32][piece hi/mid/lo][8-channel group 0..3][feature row 0..127][8 channels], zero rows beyond N."""
    n, k = weight.shape
    nt, nc = (n + 127) // 128, k // 32
    w = torch.zeros((nt * 128, k), dtype=torch.float32, device=weight.device)
    w[:n] = weight.detach().float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    r1 = torch.where(torch.isfinite(r1), r1, torch.zeros_like(r1))
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    pieces = torch.stack([hi, mid, lo], dim=0).view(torch.int16)           # [3][ntile*128][K]
    pieces = pieces.reshape(3, nt, 128, nc, 4, 8)                          # piece, tile, row, chunk, group, j
    return pieces.permute(1, 3, 0, 4, 2, 5).contiguous()                   # tile, chunk, piece, group, row, j


def _packed_linear(weight):
    ver = (weight._version, weight.data_ptr(), str(weight.device))
    hit = getattr(weight, "_ct_lin_split", None)
    if hit is None or hit[0] != ver:
        hit = (ver, pack_linear_weight_split(weight))
        weight._ct_lin_split = hit
    return hit[1]


def pack_linear_weight_ws16(weight):
    """Linear weight [N, K] -> ct_linear_ws16_f32 operand: slices of 256 input channels x 128 output features as fp16 (hi, lo) bit
    patterns (int16) [slice][piece][k step 0..15][lane half 0..1][feature 0..127][8 channels] of weight * 2^w_exp, channel of
    (k step s, half h, j) = 128 h + 8 s + j within the slice; K == 256: N / 128 feature slices, else (N == 128): K / 256 channel
    slices.  Returns (image, w_exp); the largest |weight| lands in [2^11, 2^12)."""
    n, k = weight.shape
    w = weight.detach().float()
    amax = float(w.abs().max())
    w_exp = 0 if not (amax > 0 and amax < float("inf")) else 12 - (int(np.floor(np.log2(amax))) + 1)
    w_exp = max(-100, min(100, w_exp))
    ws = w * (2.0 ** w_exp)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    pieces = torch.stack([hi, lo], dim=0).view(torch.int16)                # [2][N][K]
    if k == 256 and n % 128 == 0:
        img = pieces.reshape(2, n // 128, 128, 2, 16, 8).permute(1, 0, 4, 3, 2, 5)       # slice, piece, s, h, f, j
    elif n == 128 and k % 256 == 0:
        img = pieces.reshape(2, 128, k // 256, 2, 16, 8).permute(2, 0, 4, 3, 1, 5)
    elif k == 128 and n % 128 == 0:                                        # 128-channel slices: channel = 64 h + 8 s + j, s < 8
        img = pieces.reshape(2, n // 128, 128, 2, 8, 8).permute(1, 0, 4, 3, 2, 5)
    else:
        raise CtHipError("pack_linear_weight_ws16: K in (128, 256) with N % 128 == 0, or N == 128 with K % 256 == 0")
    return img.contiguous(), w_exp


def _packed_linear_ws16(weight):
    ver = (weight._version, weight.data_ptr(), str(weight.device))
    hit = getattr(weight, "_ct_lin_ws16", None)
    if hit is None or hit[0] != ver:
        hit = (ver, pack_linear_weight_ws16(weight))
        weight._ct_lin_ws16 = hit
    return hit[1]


_lin_ws16 = os.environ.get("CT_HIP_LINEAR_WS16", "1") != "0"


def linear_tokens_multi(x, weights, biases=None, mode=None):
    """[linear_tokens(x, w, b) for w, b in zip(weights, biases)] for up to four 128 -> 128 layers reading the SAME tokens (the q / k /
    v projections of a transformer layer, transformer.py:26-31): one launch of ct_linear_ws16_f32 whose feature slices share the
    tokens through L2 and write one result slab each.  Falls back to separate calls where that kernel does not apply."""
    biases = list(biases) if biases is not None else [None] * len(weights)
    k = x.shape[-1]
    t = x.numel() // k
    ok = ((mode or _conv_mode) == "split" and _lin_ws16 and k == 128 and 1 <= len(weights) <= 4 and t >= 4096 and
          all(tuple(w.shape) == (128, 128) for w in weights) and (all(b is None for b in biases) or all(b is not None for b in biases)))
    if not ok:
        return [linear_tokens(x, w, b, mode=mode) for w, b in zip(weights, biases)]
    _f32c(x, *weights, *biases)
    ver = tuple((w._version, w.data_ptr()) for w in weights) + (str(x.device),)
    hit = getattr(weights[0], "_ct_lin_ws16_multi", None)
    if hit is None or hit[0] != ver:
        hit = (ver, pack_linear_weight_ws16(torch.cat([w.detach() for w in weights], dim=0)))
        weights[0]._ct_lin_ws16_multi = hit
    img, w_exp = hit[1]
    bias = torch.cat([b.detach().float() for b in biases]) if biases[0] is not None else None
    out = torch.empty((len(weights),) + tuple(x.shape[:-1]) + (128,), dtype=torch.float32, device=x.device)
    check(lib().ct_linear_ws16_f32(_ptr(x), _c_p(0), 128, _ptr(img), int(w_exp), _opt(bias), _ptr(out), t, 128, 128 * len(weights), 0,
                                   _c_p(0), _c_p(0), _c_p(0), _stream()))
    return [out[i] for i in range(len(weights))]


def linear_layernorm128(x, weight, bias, gamma, beta, residual=None, mode=None):
    """[residual +] LayerNorm_128(linear(x, weight, bias)) for a 128 -> 128 layer (the merge projection with norm1 and the skip,
    transformer.py:120-127,139-147): one launch of ct_linear_ws16_f32 where that kernel applies, else the two kernels"""
    t = x.numel() // x.shape[-1]
    if ((mode or _conv_mode) == "split" and _lin_ws16 and tuple(weight.shape) == (128, 128) and x.shape[-1] == 128 and t >= 4096):
        _f32c(x, weight, bias, gamma, beta, residual)
        img, w_exp = _packed_linear_ws16(weight)
        out = torch.empty_like(x)
        check(lib().ct_linear_ws16_f32(_ptr(x), _c_p(0), 128, _ptr(img), int(w_exp), _opt(bias), _ptr(out), t, 128, 128, 0,
                                       _ptr(gamma), _ptr(beta), _opt(residual), _stream()))
        return out
    return layernorm128(linear_tokens(x, weight, bias, mode=mode), gamma, beta, residual=residual)


def set_linear_ws16(on):
    """True (default): the FFN-shaped linears (K = 256 -> N % 128 = 0; K % 256 = 0 -> N = 128 as partial slabs) run in
    ct_linear_ws16_f32 (weight slice resident in LDS, two fp16 pieces); False: everything in ct_linear_tokens_split_f32"""
    global _lin_ws16
    _lin_ws16 = bool(on)


def linear_tokens(x, weight, bias=None, act=ACT_NONE, x2=None, mode=None, partials=False):
    """x [..., K1] channels-last tokens (optionally concatenated with x2 [..., K2] on the fly), weight [N, K1+K2]
    (PyTorch layout) -> [..., N].  mode (default: conv_mode()): "split" = 16-bit matrix pipe, float32-grade (K % 32 == 0; the
    pre-split weight is cached on the weight tensor); "exact" = v_mfma_f32_32x32x2_f32.
    partials=True: the result may come back as [P, ..., N] slabs whose sum over P is the result (K-sliced ct_linear_ws16_f32;
    layernorm128 adds them on its way in) -- the caller must accept either shape (P == 1 slab otherwise)."""
    _f32c(x, weight, bias, x2)
    k1, n = x.shape[-1], weight.shape[0]
    k = k1 + (x2.shape[-1] if x2 is not None else 0)
    if weight.shape[1] != k or (x2 is not None and x2.shape[:-1] != x.shape[:-1]):
        raise CtHipError("linear_tokens: shape mismatch")
    t = x.numel() // k1
    if (mode or _conv_mode) == "split" and _lin_ws16 and act in (ACT_NONE, ACT_GELU) and t >= 4096:
        nsl = k == 256 and n % 128 == 0 and n // 128 in (1, 2, 4, 8) and (k1 == 128 if x2 is not None else True)
        ksl = (not nsl) and partials and n == 128 and k % 256 == 0 and k // 256 in (2, 4, 8) and x2 is None and act == ACT_NONE
        if nsl or ksl:
            img, w_exp = _packed_linear_ws16(weight)
            shape = ((k // 256,) if ksl else ()) + tuple(x.shape[:-1]) + (n,)
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
            check(lib().ct_linear_ws16_f32(_ptr(x), _opt(x2), k1, _ptr(img), int(w_exp), _opt(bias), _ptr(out), t, k, n, int(act),
                                           _c_p(0), _c_p(0), _c_p(0), _stream()))
            return out
    out = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    if (mode or _conv_mode) == "split" and k % 32 == 0:
        check(lib().ct_linear_tokens_split_f32(_ptr(x), _opt(x2), k1, _ptr(_packed_linear(weight)), _opt(bias), _ptr(out), t, k, n,
                                               int(act), _stream()))
    else:
        check(lib().ct_linear_tokens_f32(_ptr(x), _opt(x2), k1, _ptr(weight), _opt(bias), _ptr(out), t, k, n, int(act), _stream()))
    return out


def layernorm128(x, gamma, beta, residual=None, partials=1):
    """LayerNorm(128) (+ residual) of x [..., 128]; partials = P > 1: x is [P, ..., 128] and the input is the sum of its P slabs
    (added in slab order: the K-sliced linear's partial results)"""
    _f32c(x, gamma, beta, residual)
    if x.shape[-1] != 128:
        raise CtHipError("layernorm128: last dim must be 128")
    partials = int(partials)
    if partials < 1 or (partials > 1 and x.shape[0] != partials):
        raise CtHipError("layernorm128: x must be [partials, ..., 128]")
    out = torch.empty(x.shape[1:] if partials > 1 else x.shape, dtype=torch.float32, device=x.device)
    check(lib().ct_layernorm128_f32(_ptr(x), _ptr(gamma), _ptr(beta), _opt(residual), _ptr(out), out.numel() // 128, partials, _stream()))
    return out


def attention_tokens(q, k, v, region=None, scale=None, rowmap=None, nsplit=None, kv_shift=0):
    """q,k [B,L,128], v [B,L,128] or [B,L,2]; region int32 [B,L] or None -> [B,L,cv].
    With rowmap (int32 [B', L']): B' x L' attention problems whose token (b, i) is row rowmap[b, i] of the flattened
    q / k / v / out -- window partitions without copies; the result has v's shape.  kv_shift (with rowmap): keys / values are
    read kv_shift rows further (mod the row count) than the queries: cross attention to the other half of the batch."""
    _f32c(q, k, v)
    c, cv = q.shape[-1], v.shape[-1]
    for t in (region, rowmap):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise CtHipError("region / rowmap must be contiguous int32")
    if rowmap is not None:
        b, l = rowmap.shape
        if region is not None and tuple(region.shape) != (b, l):
            raise CtHipError("region and rowmap must have the same shape")
        if b * l != q.numel() // c or b * l != v.numel() // cv or k.shape != q.shape:
            raise CtHipError("rowmap must be a permutation of the token rows")
        out = torch.empty_like(v)
    else:
        b, l, _ = q.shape
        out = torch.empty((b, l, cv), dtype=torch.float32, device=q.device)
    if nsplit is None:
        # key split so that ~2 workgroups per CU exist (global matching at 1/8 resolution launches only 56 otherwise)
        wgs = b * ((l + 127) // 128)
        nsplit = 1 if wgs >= 384 else max(1, min(8, 512 // max(wgs, 1), (l + 255) // 256))
    ws, need = None, 0
    if nsplit > 1:
        need = lib().ct_attention_workspace_bytes(b, l, cv, nsplit)
        ws = workspace(-2, 0, 0, q.device, need=need)
    check(lib().ct_attention_tokens_f32(_ptr(q), _ptr(k), _ptr(v), _opt(region), _opt(rowmap), _ptr(out), b, l, cv,
                                        float(scale if scale is not None else c ** -0.5), nsplit, _opt(ws), need, int(kv_shift), _stream()))
    return out


def local_corr_softmax(f0, f1, h, w, radius):
    _f32c(f0, f1)
    b = f0.shape[0]
    flow = torch.empty((b, 2, h, w), dtype=torch.float32, device=f0.device)
    check(lib().ct_local_corr_softmax_f32(_ptr(f0), _ptr(f1), _ptr(flow), b, h, w, radius, _stream()))
    return flow


def local_corr_flow(f0, f1, flow, radius):
    _f32c(f0, f1, flow)
    b, _, h, w = flow.shape
    corr = torch.empty((b, (2 * radius + 1) ** 2, h, w), dtype=torch.float32, device=f0.device)
    check(lib().ct_local_corr_flow_f32(_ptr(f0), _ptr(f1), _ptr(flow), _ptr(corr), b, h, w, radius, _stream()))
    return corr


def local_attn_prop(q, k, flow, radius):
    _f32c(q, k, flow)
    b, _, h, w = flow.shape
    out = torch.empty_like(flow)
    check(lib().ct_local_attn_prop_f32(_ptr(q), _ptr(k), _ptr(flow), _ptr(out), b, h, w, radius, _stream()))
    return out


def bilinear_resize(x, size, mul0=1.0, mul1=1.0):
    _f32c(x)
    n, c, h, w = x.shape
    out = torch.empty((n, c, size[0], size[1]), dtype=torch.float32, device=x.device)
    check(lib().ct_bilinear_resize_f32(_ptr(x), _ptr(out), n, c, h, w, size[0], size[1], float(mul0), float(mul1), _stream()))
    return out


def flow_warp(img, flow):
    _f32c(img, flow)
    n, c, h, w = img.shape
    out = torch.empty_like(img)
    check(lib().ct_flow_warp_f32(_ptr(img), _ptr(flow), _ptr(out), n, c, h, w, _stream()))
    return out


def convex_upsample(flow, mask, factor):
    _f32c(flow, mask)
    b, _, h, w = flow.shape
    out = torch.empty((b, 2, h * factor, w * factor), dtype=torch.float32, device=flow.device)
    check(lib().ct_convex_upsample_f32(_ptr(flow), _ptr(mask), _ptr(out), b, h, w, factor, _stream()))
    return out


def fb_check(fwd, bwd, alpha=0.01, beta=0.5):
    """forward_backward_consistency_check (geometry.py:78-99) -> (fwd_occ, bwd_occ) [B,H,W] as 0/1 floats"""
    _f32c(fwd, bwd)
    b, _, h, w = fwd.shape
    wb, wf = flow_warp(bwd, fwd), flow_warp(fwd, bwd)
    fo = torch.empty((b, h, w), dtype=torch.float32, device=fwd.device)
    bo = torch.empty_like(fo)
    check(lib().ct_fb_check_f32(_ptr(fwd), _ptr(bwd), _ptr(wb), _ptr(wf), _ptr(fo), _ptr(bo), b, h, w, alpha, beta, _stream()))
    return fo, bo


def nchw_to_tokens(x):
    """[B,C,H,W] -> channels-last tokens [B, H*W, C] (transformer.py:238-239's flatten + permute) through the LDS-tiled transpose"""
    _f32c(x)
    b, c, h, w = x.shape
    out = torch.empty((b, h * w, c), dtype=torch.float32, device=x.device)
    check(lib().ct_nchw_to_rows_f32(_ptr(x), _ptr(out), b, c, h, w, c * h * w, c, 0, _stream()))
    return out


def tokens_to_nchw(t, h, w):
    """tokens [B, H*W, C] -> [B,C,H,W]"""
    _f32c(t)
    b, l, c = t.shape
    if l != h * w:
        raise CtHipError("tokens_to_nchw: %d tokens are not %d x %d" % (l, h, w))
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=t.device)
    check(lib().ct_rows_to_nchw_f32(_ptr(t), _ptr(out), b, c, h, w, c * h * w, c, 0, _stream()))
    return out


def pam_streaming(q, k, v, rgb, q_other, k_other):
    """DCMCS3DI's parallax attention through the streaming kernels (any width):
    q,k [B,64,H,W] = Q(left), K(right); v [B,64,H,W], rgb [B,3,H,W]; q_other,k_other = Q(right), K(left).
    Returns (fea_warped [B,64,H,W], warped_rgb [B,3,H,W], valid [B,1,H,W] 0/1, colsum [B,1,H,W])."""
    _f32c(q, k, v, rgb, q_other, k_other)
    b, c, h, w = q.shape
    if c != 64 or v.shape[1] != 64:
        raise CtHipError("pam_streaming is built for 64 channels")
    vt = torch.empty((b * h, w, 96), dtype=torch.float32, device=q.device)
    nchw_to_rows(v, vt, 0)
    return pam_streaming_rows(nchw_to_rows(q), nchw_to_rows(k), vt, rgb, nchw_to_rows(q_other), nchw_to_rows(k_other))


def nchw_to_rows(t, out=None, c0=0):
    """[B,C,H,W] -> channels c0.. of a [B*H, W, C'] token-rows tensor (data movement only)"""
    b, ct, h, w = t.shape
    if out is None:
        out = torch.empty((b * h, w, ct), dtype=torch.float32, device=t.device)
    check(lib().ct_nchw_to_rows_f32(_ptr(t), _ptr(out), b, ct, h, w, _nchw_bstride(t), out.shape[2], c0, _stream()))
    return out


def pam_streaming_rows(qt, kt, vt, rgb, qo, ko):
    """pam_streaming on token rows: qt, kt, qo, ko [B*H, W, 64] (contiguous; views of a larger rows tensor along dim 0 are fine);
    vt [B*H, W, 96] with the value in channels 0..63 -- channels 64..95 are filled here (rgb [B,3,H,W] + zero padding)."""
    b, _, h, w = rgb.shape
    for t in (qt, kt, qo, ko):
        if t.shape != (b * h, w, 64) or not t.is_contiguous() or t.dtype != torch.float32 or not t.is_cuda:
            raise CtHipError("pam_streaming_rows needs contiguous float32 [B*H, W, 64] CUDA tensors")
    if vt.shape != (b * h, w, 96) or not vt.is_contiguous() or vt.dtype != torch.float32:
        raise CtHipError("pam_streaming_rows needs a contiguous float32 [B*H, W, 96] value tensor")
    _f32c(rgb)
    scale = 1.0 / 64                                  # the reference scales by 1/c, not 1/sqrt(c) (attention.py:41)

    def nchw(t, ct, c0):                              # [B*H, W, C'] tokens -> [B,ct,H,W] from channels c0..c0+ct
        out = torch.empty((b, ct, h, w), dtype=torch.float32, device=t.device)
        check(lib().ct_rows_to_nchw_f32(_ptr(t), _ptr(out), b, ct, h, w, ct * h * w, t.shape[2], c0, _stream()))
        return out
    vt[:, :, 67:] = 0.0                               # the 29 padding channels of the 96-channel value
    nchw_to_rows(rgb, vt, 64)
    out = torch.empty((b * h, w, 96), dtype=torch.float32, device=qt.device)
    check(lib().ct_attention_rows64_f32(_ptr(qt), _ptr(kt), _ptr(vt), _ptr(out), _c_p(0), b * h, w, scale, _stream()))
    fea = nchw(out, 64, 0)
    wrgb = nchw(out, 3, 64)
    stats = torch.empty((b * h, w, 2), dtype=torch.float32, device=qt.device)
    check(lib().ct_attention_rows64_f32(_ptr(qo), _ptr(ko), _c_p(0), _c_p(0), _ptr(stats), b * h, w, scale, _stream()))
    colsum = torch.empty((b * h, w), dtype=torch.float32, device=qt.device)
    check(lib().ct_attention_colsum64_f32(_ptr(qo), _ptr(ko), _ptr(stats), _ptr(colsum), b * h, w, scale, _stream()))
    colsum = colsum.view(b, 1, h, w)
    valid = (colsum > 0.1).float()                    # threshold only (utils.py:34); the sums come from the kernel
    return fea, wrgb, valid, colsum
