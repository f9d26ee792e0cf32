"""CPU-only checks: the C-ABI library loads and exports every symbol include/ct_hip.h declares, the
ctypes table matches the header, and the host-side logic (dtype rules, rotation drawing, packing,
sharding arithmetic) behaves.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ct_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ct_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import ct_hip
    assert os.path.exists(ct_hip.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(ct_hip.LIB_PATH)
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    # the ctypes signature table binds exactly the declared entry points
    assert sorted(ct_hip.SIGNATURES.keys()) == declared
    assert ct_hip.lib().ct_abi_version() == ct_hip.CT_ABI_VERSION == 9
    assert re.search(r"#define CT_ABI_VERSION 9\b", open(HEADER).read())
    assert ct_hip.lib().ct_error_string(-2).decode().startswith("workspace")
    assert ct_hip.lib().ct_workspace_bytes(ct_hip.CT_WS_REINHARD, 1920 * 1080, 4) > 0
    assert ct_hip.lib().ct_idt_workspace_bytes(1, 4, 255) > 0
    assert ct_hip.lib().ct_idt_workspace_bytes(1, 4, 4096) == 0          # bins above the LDS budget are refused


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ct_hip
    import methods.linear as lin
    import methods.iterative as it
    x = np.random.default_rng(0).random((4, 4, 3)).astype(np.float32)
    with pytest.raises(ct_hip.CtHipError):
        lin.color_transfer_between_images(x, x)
    with pytest.raises(ct_hip.CtHipError):
        lin.monge_kantorovitch_color_transfer(x, x)
    with pytest.raises(ct_hip.CtHipError):
        it.iterative_distribution_transfer(x, x)
    with pytest.raises(ct_hip.CtHipError):
        ct_hip.lab_stats(torch.zeros(2, 2, 3))
    from methods.dcmcs3di import DCMCS3DI
    m = DCMCS3DI(extraction_layers=1, transfer_layers=1)
    with pytest.raises(ct_hip.CtHipError):
        m(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8), inference=True)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under color-transfer_amd/ may import, include or link it."""
    pkg = os.path.join(ROOT, "color-transfer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                assert not re.search(r"^\s*(import|from)\s+oracle\b", open(path).read(), flags=re.M), path
            elif f.endswith((".hip", ".h", ".c", ".cpp")):
                includes = [l for l in open(path).read().splitlines() if l.lstrip().startswith("#include")]
                assert not any("oracle" in l for l in includes), path
            elif f == "Makefile":
                code = [l for l in open(path).read().splitlines() if not l.lstrip().startswith("#")]
                assert not any("oracle" in l for l in code), path


def test_host_rules_linear():
    import methods.linear as lin
    with pytest.raises(ValueError):
        lin.monge_kantorovitch_color_transfer(np.zeros((2, 2, 3)), np.zeros((2, 2, 3)), decomposition="nope")
    with pytest.raises(ValueError):
        lin._as_float(np.zeros((2, 2, 4)))
    assert lin._as_float(np.zeros((2, 2, 3), np.uint8)).dtype == np.float64
    assert lin._as_float(np.zeros((2, 2, 3), np.float32)).dtype == np.float32
    # the 3x3 algebra is the reference's (same numpy/scipy calls): check it against the oracle's copy
    from oracle import linear as olin
    rng = np.random.default_rng(1)
    a, b = rng.random((50, 3)), rng.random((60, 3)) * 0.5
    ca, cb = np.cov(a.T), np.cov(b.T)
    np.testing.assert_allclose(lin.xiao_matrix(ca, cb), olin.xiao_matrix(ca, cb), rtol=0, atol=0)
    for d in ("MK", "sqrt", "cholesky"):
        np.testing.assert_allclose(lin.mk_matrix(ca, cb, d), olin.mk_matrix(ca, cb, d), rtol=0, atol=0)
    out = lin.color_transfer_between_images(np.zeros((0, 5, 3), np.float32), np.zeros((2, 2, 3), np.float32))
    assert out.shape == (0, 5, 3)


def test_dtype_rules_follow_the_reference():
    """only Reinhard goes through img_as_float (skimage rgb2lab); Xiao / MK / IDT see integer frames on their own scale"""
    import methods.linear as lin
    u8 = np.arange(24, dtype=np.uint8).reshape(2, 4, 3)
    assert lin._as_float(u8).max() == 23 / 255.0 and lin._as_raw_float(u8).max() == 23.0
    assert lin._as_raw_float(u8).dtype == np.float64 and lin._as_raw_float(u8.astype(np.float32)).dtype == np.float32
    with pytest.raises(ValueError):
        lin._as_raw_float(np.zeros((2, 2, 4)))
    out = lin.color_transfer_between_images(np.zeros((0, 5, 3), np.float32), np.zeros((2, 2, 3), np.float64))
    assert out.dtype == np.float64                    # numpy promotion of the reference's Lab arithmetic


def test_cli_override_syntax():
    from utils import cli
    cfg = {"model": {"class_path": "methods.Runner", "init_args": {"func_spec": "a.b"}}, "data": {"init_args": {"num_workers": 0}}}
    cli._set(cfg, "model.func_spec", "x.y")
    assert cfg["model"]["init_args"]["func_spec"] == "x.y"
    cli._set(cfg, "model.init_args.func_spec", "p.q")                # LightningCLI's canonical spelling
    assert cfg["model"]["init_args"] == {"func_spec": "p.q"}
    cli._set(cfg, "data.init_args.n_frames", "5")
    cli._set(cfg, "data.height", "64")
    assert cfg["data"]["init_args"] == {"num_workers": 0, "n_frames": 5, "height": 64}
    cli._set(cfg, "model.class_path", "methods.dcmcs3di.DCMCS3DI")
    assert cfg["model"]["class_path"] == "methods.dcmcs3di.DCMCS3DI" and "class_path" not in cfg["model"]["init_args"]
    cli._set(cfg, "trainer.logger", "false")
    assert cfg["trainer"] == {"logger": False}
    with pytest.raises(SystemExit):
        cli.main(["test", "--config"])                                 # dangling flag
    with pytest.raises(SystemExit):
        cli.main(["fit"])


def test_rotations_follow_numpy_global_rng():
    import scipy.stats
    import methods.iterative as it
    np.random.seed(11)
    want = np.stack([scipy.stats.special_ortho_group.rvs(3) for _ in range(4)])
    np.random.seed(11)
    got = it.draw_rotations(4)
    assert np.array_equal(want, got)
    assert np.array_equal(it.draw_rotations(4, seed=11), want)
    x = np.random.default_rng(0).random((3, 3, 3)).astype(np.float32)
    assert it.iterative_distribution_transfer(x, x, n_iter=0) is not None   # n_iter=0 never touches the GPU


def test_conv_weight_packing_layout():
    import ct_hip
    w = torch.arange(5 * 3 * 3 * 3, dtype=torch.float32).reshape(5, 3, 3, 3)
    b = torch.arange(5, dtype=torch.float32)
    wp, bp = ct_hip.pack_conv_weight(w, b)
    assert wp.shape == (9, 2, 2, 32) and bp.shape == (32,)
    for (co, ci, ky, kx) in [(0, 0, 0, 0), (4, 2, 2, 1), (3, 1, 1, 1)]:
        assert wp[ky * 3 + kx, ci // 2, ci % 2, co] == w[co, ci, ky, kx]
    assert wp[:, 1, 1].abs().sum() == 0 and wp[..., 5:].abs().sum() == 0      # zero padding (cin 3 -> 4, cout 5 -> 32)
    assert torch.equal(bp[:5], b) and bp[5:].abs().sum() == 0


def test_fp16_piece_packings_layout_and_value():
    """the host packings of the two-piece fp16 kernels: hi + lo == weight * 2^w_exp to 2^-22 relative, the largest weight in
    [2^11, 2^12), and every (slice, piece, step, half, feature, channel) index where include/ct_hip.h says it is"""
    import ct_hip
    g = torch.Generator().manual_seed(3)

    def pieces_value(img):                              # int16 bit patterns [..., piece axis first of the two given] -> float64
        return img.view(torch.float16).double()
    # linear, feature slices of 256 channels (FFN1), channel slices (FFN2), 128-channel slices (q / k / v stacked)
    for n, k in ((256, 256), (128, 512), (384, 128)):
        w = torch.randn(n, k, generator=g) * 0.03
        img, w_exp = ct_hip.pack_linear_weight_ws16(w)
        amax = (w.abs().max() * 2.0 ** w_exp).item()
        assert 2048 <= amax < 4096
        v = pieces_value(img)
        steps = 16 if k != 128 else 8
        half = 8 * steps
        if k == 256 or k == 128:
            assert img.shape == (n // 128, 2, steps, 2, 128, 8)
            for (sl, s, h, f, j) in [(0, 0, 0, 0, 0), (n // 128 - 1, steps - 1, 1, 127, 7), (0, 3, 1, 17, 5)]:
                want = w[128 * sl + f, half * h + 8 * s + j].double() * 2.0 ** w_exp
                got = v[sl, 0, s, h, f, j] + v[sl, 1, s, h, f, j]
                assert abs(got - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -24
        else:
            assert img.shape == (k // 256, 2, 16, 2, 128, 8)
            for (sl, s, h, f, j) in [(0, 0, 0, 0, 0), (1, 15, 1, 127, 7), (1, 2, 0, 9, 3)]:
                want = w[f, 256 * sl + 128 * h + 8 * s + j].double() * 2.0 ** w_exp
                got = v[sl, 0, s, h, f, j] + v[sl, 1, s, h, f, j]
                assert abs(got - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -24
    # convolution, any tap count: [group][chunk16][tap][piece][m][k-half][cout % 32][8 channels]
    for (cout, cin, kh, kw) in ((70, 40, 1, 5), (64, 64, 3, 3), (5, 3, 1, 1)):
        w = torch.randn(cout, cin, kh, kw, generator=g) * 1e-3
        img, w_exp = ct_hip.pack_conv_weight_split16(w)
        assert img.shape == ((cout + 63) // 64, (cin + 15) // 16, kh * kw, 2, 2, 2, 32, 8)
        v = pieces_value(img)
        for (co, ci, ky, kx) in [(0, 0, 0, 0), (cout - 1, cin - 1, kh - 1, kw - 1), (cout // 2, cin // 3, 0, kw // 2)]:
            want = w[co, ci, ky, kx].double() * 2.0 ** w_exp
            idx = (co // 64, ci // 16, ky * kw + kx, slice(None), (co % 64) // 32, (ci % 16) // 8, co % 32, ci % 8)
            got = v[idx].sum()
            assert abs(got - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -24
        assert v[(cout + 63) // 64 - 1, :, :, :, 1, :, (cout % 32 or 32):].abs().sum() == 0 or cout % 64 == 0 or cout % 64 > 32


def test_winograd_weight_packing_layout_and_value():
    """ct_hip.pack_conv_weight_wino16 (the operand of ct_conv3x3_wino16_f32, include/ct_hip.h): hi + lo == (G g G^T) * 2^w_exp to
    2^-21 relative with G g G^T taken in float64, the largest element in [2^11, 2^12), every (group, position, cout block, cin
    chunk, piece, lane, element) where the header says it is, zero padding of absent channels"""
    import ct_hip
    g = torch.Generator().manual_seed(5)
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    for cout, cin in ((64, 64), (70, 40), (5, 33)):
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        img, w_exp = ct_hip.pack_conv_weight_wino16(w)
        groups = (cout + 63) // 64
        assert img.shape == (groups, 16, 4, 2, 2, 64, 8) and img.dtype == torch.int16
        u = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G)             # [cout][cin][4][4]
        amax = (u.float().abs().max() * 2.0 ** w_exp).item()
        assert 2048 <= amax < 4096
        v = img.view(torch.float16).double()
        for (co, ci, pi, pj) in [(0, 0, 0, 0), (cout - 1, cin - 1, 3, 3), (cout // 2, cin // 3, 1, 2), (cout - 1, 0, 2, 1)]:
            want = float(u[co, ci, pi, pj].float()) * 2.0 ** w_exp        # the transform is rounded once to float32
            lane = 16 * ((ci % 32) // 8) + (co % 64) % 16
            idx = (co // 64, 4 * pi + pj, (co % 64) // 16, ci // 32, slice(None), lane, ci % 8)
            got = v[idx].sum().item()
            assert abs(got - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -24, (co, ci, pi, pj, got, want)
        if cin < 64:                                                        # channels cin .. 63 are zero
            lanes = [16 * ((c % 32) // 8) + m for c in range(cin, 64) for m in range(16) if c % 8 == 0]
            assert v[:, :, :, 1, :, :, :][..., [16 * 3 + m for m in range(16)], 7].abs().sum() == 0      # cin 63: chunk 1, k-block 3, element 7
        if cout % 64:                                                       # couts past cout are zero
            co = cout % 64
            assert v[groups - 1, :, co // 16, :, :, [16 * kb + co % 16 for kb in range(4)], :].abs().sum() == 0


def test_sharding_arithmetic():
    from utils import sharding as sh
    assert sh.frames_of_rank(10, 1, 4) == [1, 5, 9]
    assert sum(len(sh.frames_of_rank(1000, r, 8)) for r in range(8)) == 1000
    assert sh.padded_local_count(10, 4) == 3
    assert sh.frame_seed(7) == 1241
    # single process: identity
    m = torch.arange(12, dtype=torch.float64).reshape(6, 2)
    assert torch.equal(sh.gather_frame_metrics(m, 6, rank=0, world=1), m)


def test_persistent_reinhard_entries_reject_bad_arguments_without_a_gpu():
    """argument checks of the round-4 entries return before anything touches the device"""
    import ct_hip
    lib = ct_hip.lib()
    assert lib.ct_reinhard_persist_supported(255) == 0 and lib.ct_reinhard_persist_supported(1920 * 1080) == 1
    assert lib.ct_reinhard_persist_supported(64 * 1024 * 1024) == 0
    assert lib.ct_workspace_bytes(ct_hip.CT_WS_REINHARD_PERSIST, 255, 4) == 0
    assert lib.ct_workspace_bytes(ct_hip.CT_WS_REINHARD_PERSIST, 1920 * 1080, 4) > 0
    assert lib.ct_workspace_bytes(ct_hip.CT_WS_REINHARD, 1920 * 1080, 4) >= lib.ct_workspace_bytes(ct_hip.CT_WS_REINHARD_PERSIST, 1920 * 1080, 4)
    null = None
    for fn in (lib.ct_reinhard_persist_f32, lib.ct_reinhard_psnr_u8):
        assert fn(null, null, null, null, null, 1920 * 1080, 1, null, null, 0, null) == -1        # null images
        assert fn(null, null, null, null, null, 1920 * 1080, 0, null, null, 0, null) == 0         # an empty batch is a no-op
        assert fn(null, null, null, null, null, -1, 1, null, null, 0, null) == -1
    assert lib.ct_fft2d_c2c_f32(null, 4, 4, 1, 0, null) == -1
    assert lib.ct_set_lab_mode_thread(7) == -1 and lib.ct_set_lab_mode_thread(-1) == 0


def test_conv_scratch_argument_checks_without_a_gpu():
    """ct_conv2d_split_f32 checks its stream-K scratch (alignment, size) before anything touches the device"""
    import ctypes
    import ct_hip
    lib = ct_hip.lib()
    need = lib.ct_conv_split_scratch_bytes()
    assert need >= 4096 + 512 * 256 * 64 * 4 and need % 16 == 0
    host = (ctypes.c_char * 256)()
    p = ctypes.cast(host, ctypes.c_void_p)
    addr = p.value

    def call(scratch, nbytes, n=0):
        return lib.ct_conv2d_split_f32(p, None, 0, None, 0, p, p, None, p, n, 16, 64, 8, 32, 3, 3, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, None, None,
                                       scratch, nbytes, None)
    aligned = ctypes.c_void_p((addr + 15) & ~15)
    assert call(None, 0) == 0                                           # an empty batch, no scratch: a no-op
    assert call(aligned, need) == 0                                     # ... with a well-formed scratch too
    assert call(aligned, need - 1) == -2                                # CT_E_WORKSPACE: too small
    assert call(ctypes.c_void_p(aligned.value + 4), need) == -2         # misaligned


def test_first_library_use_through_any_entry_cannot_deadlock():
    """Round 5 regression: the stream-K scratch helper took the binding's lock and then called lib(), which takes it on first
    use -- a process whose FIRST library call was a tile convolution hung forever (tools/bench_dcmcs3di.py under rocprofv3; bench.py
    and the tests load the library earlier and never saw it).  Fresh interpreter, no GPU: the call must come back (with the error of
    the missing device), not hang."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import torch, ct_hip, types\n"
            "torch.cuda.current_stream = lambda d=None: types.SimpleNamespace(cuda_stream=0)\n"
            "torch.cuda.is_current_stream_capturing = lambda: False\n"
            "try:\n    ct_hip._conv_scratch(torch.device('cuda', 0))\nexcept Exception as e:\n    print('raised', type(e).__name__)\n"
            "print('lib loaded', ct_hip._lib is not None)\n") % (ROOT, os.path.join(ROOT, "color-transfer_amd"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "lib loaded True" in p.stdout, (p.stdout, p.stderr[-500:])
