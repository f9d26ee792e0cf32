"""GPU parity of the weight-stationary 3x3 convolution (csrc/conv_ws.hip) in its fp16 two-piece form (ct_conv3x3_ws16_f32) and
its bf16 three-piece form, against float64 torch convolutions: every geometry branch (partial strips, row segments that are
not multiples of three, fewer than four input chunks, odd channel counts, several output groups), every epilogue, and the
per-row power-of-two scaling (rows of very different magnitude, zero rows, scale changes between neighbouring rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F   # noqa: E402

G = torch.Generator().manual_seed(11)


def rnd(*shape):
    return torch.randn(*shape, generator=G)


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


@pytest.fixture(autouse=True)
def direct_kernel(hip):
    """This module tests the direct kernels (conv_ws, conv_split) by name: the Winograd form that takes conv_ws's convolutions by
    default is switched off here and on again inside its own tests (test_conv_wino_*)."""
    hip.set_conv_wino(False)
    yield
    hip.set_conv_wino(True)


CASES = [(2, 64, 64, 24, 64), (1, 64, 32, 9, 36), (1, 48, 64, 50, 100), (1, 33, 7, 5, 8), (1, 64, 130, 40, 64), (3, 64, 64, 17, 32)]


@pytest.mark.parametrize("ws16", [True, False])
@pytest.mark.parametrize("cfg", CASES)
def test_conv_ws_vs_float64(hip, cfg, ws16):
    n, cin, cout, h, w = cfg
    x, wt, b = rnd(n, cin, h, w) * 3, rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5, rnd(cout)
    res = rnd(n, cout, h, w)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    assert wp._ct_split[2] is not None                      # the fp16 image travels with the packing
    tol = 2e-6 * max(1.0, ref.abs().max().item())
    hip.set_conv_ws16(ws16)
    try:
        for act, fn in ((0, lambda t: t), (1, lambda t: F.leaky_relu(t, 0.01))):
            out = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=act)
            assert (out.double().cpu() - fn(ref)).abs().max().item() < tol, (cfg, ws16, act)
        out = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=1, residual=res.cuda(), clamp=True)
        assert (out.double().cpu() - (F.leaky_relu(ref, 0.01) + res.double()).clamp(0, 1)).abs().max().item() < tol
        # the generic activations go through ct_conv2d_split_f32 / gconv2d's path (ReLU, sigmoid)
        wpg, bpg = hip.pack_gconv_weight(wt.cuda(), b.cuda())
        for act, fn in ((2, torch.relu), (3, torch.sigmoid)):
            out = hip.gconv2d(x.cuda(), wpg, bpg, cout, 3, 1, 1, act=act)
            assert (out.double().cpu() - fn(ref)).abs().max().item() < tol, (cfg, ws16, act)
        # non-contiguous batch views (the two views of a stereo pair live in one tensor)
        if n >= 2:
            big = rnd(n + 1, cin, h, w)
            out = hip.conv2d(big.cuda()[1:], wp, bp, cout, 3)
            want = F.conv2d(big[1:].double(), wt.double(), b.double(), padding=1)
            assert (out.double().cpu() - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
    finally:
        hip.set_conv_ws16(True)


@pytest.mark.parametrize("cfg", [(2, 64, 7, 11, 40), (3, 40, 12, 6, 36), (2, 64, 72, 9, 100)])
def test_conv_ws_writes_only_its_output(hip, cfg):
    """Output rows leave conv_ws through a buffer descriptor over a wave's eight output planes; stores that must not happen
    (channels past cout, columns past the image, rows of padding steps) get an out-of-range offset and are dropped by the
    hardware (csrc/conv_ws.hip).  The output is a view inside a larger tensor of sentinels: nothing but the view may change,
    with partial channel groups and several images (a store that was not dropped would land in the next image)."""
    n, cin, cout, h, w = cfg
    x, wt, b = rnd(n, cin, h, w), rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5, rnd(cout)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    big = torch.full((n + 2, cout, h, w), 12345.0, device="cuda")
    out = big[1:n + 1]
    hip.conv2d(x.cuda(), wp, bp, cout, 3, act=1, out=out)
    torch.cuda.synchronize()
    assert bool((big[0] == 12345.0).all()) and bool((big[n + 1] == 12345.0).all())
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.01)
    assert (out.double().cpu() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


@pytest.fixture(params=[0, 1], ids=["wino4", "wino8"])
def wino_form(hip, request):
    """both Winograd kernels behind ct_conv3x3_wino16_f32: 0 = csrc/conv_wino4.hip (four waves, pipelined; the default), 1 = csrc/conv_wino.hip"""
    hip.set_conv_wino_form(request.param)
    yield request.param
    hip.set_conv_wino_form(0)


WINO_CASES = [(2, 64, 64, 24, 64), (1, 64, 32, 9, 36), (1, 48, 64, 50, 100), (1, 33, 7, 5, 8), (2, 64, 130, 41, 64), (3, 64, 64, 17, 32),
              (1, 64, 64, 2, 4), (1, 40, 64, 1, 12)]


@pytest.mark.parametrize("cfg", WINO_CASES)
def test_conv_wino_vs_float64(hip, cfg, wino_form):
    """ct_conv3x3_wino16_f32 (csrc/conv_wino.hip: Winograd F(2x2, 3x3) on two fp16 pieces) against the float64 convolution: odd
    heights (half tile rows), widths that are not multiples of the 32-column strip, partial channel groups, several images and
    segments, activation / skip / clamp; and nothing outside the output view is written."""
    n, cin, cout, h, w = cfg
    x, wt, b = rnd(n, cin, h, w) * 3, rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5, rnd(cout)
    x[:, :, : h // 2] *= 1e-2                       # tile rows of different magnitude
    res = rnd(n, cout, h, w)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    assert wp._ct_split[3] is not None
    tol = 3e-6 * max(1.0, ref.abs().max().item())
    hip.set_conv_wino(True)
    try:
        for act, fn in ((0, lambda t: t), (1, lambda t: F.leaky_relu(t, 0.01))):
            big = torch.full((n + 2, cout, h, w), 12345.0, device="cuda")
            out = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=act, out=big[1:n + 1])
            torch.cuda.synchronize()
            assert bool((big[0] == 12345.0).all()) and bool((big[n + 1] == 12345.0).all())
            err = (out.double().cpu() - fn(ref)).abs().max().item()
            assert err < tol, (cfg, act, err, tol)
        out = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=1, residual=res.cuda(), clamp=True)
        assert (out.double().cpu() - (F.leaky_relu(ref, 0.01) + res.double()).clamp(0, 1)).abs().max().item() < tol
        a1 = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=1)
        a2 = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=1)
        assert torch.equal(a1, a2)                  # fixed summation order
    finally:
        hip.set_conv_wino(False)


def test_conv_wino_1080p_bands_vs_float64(hip, wino_form):
    """The size BASELINE.json's metric names (Missing 3): ONE 2-view 64 -> 64 launch at 1080 x 1920 with LeakyReLU and a
    skip tensor (the second convolution of a ResB, reference pasmnet/backbone.py:8-15), held against the float64 convolution on 8-row
    bands at the top, at every row-segment boundary of both kernels' decompositions (270 / 540 / 810 for the four-wave kernel) and at
    the bottom, all 1920 columns (every strip, both image edges), both views.  Rows of mixed magnitude as in the network's activations."""
    n, c, h, w = 2, 64, 1080, 1920
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, w, generator=g)
    x[:, :, 200:600] *= 1e-2
    wt, b = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5, torch.randn(c, generator=g)
    res = torch.randn(n, c, h, w, generator=g)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    hip.set_conv_wino(True)
    try:
        out = hip.conv2d(x.cuda(), wp, bp, c, 3, act=1, residual=res.cuda()).cpu()
    finally:
        hip.set_conv_wino(False)
    worst = 0.0
    for y0 in (0, 132, 266, 536, 806, 944, 1072):
        ya, yb = max(y0 - 1, 0), min(y0 + 9, h)                     # one halo row each side where the image has one
        ref = F.leaky_relu(F.conv2d(x[:, :, ya:yb].double(), wt.double(), b.double(), padding=1), 0.01)
        lo, hi = y0 - ya, y0 - ya + 8
        want = ref[:, :, lo:hi] + res[:, :, y0:y0 + 8].double()
        err = (out[:, :, y0:y0 + 8].double() - want).abs().max().item()
        tol = 3e-6 * max(1.0, ref[:, :, lo:hi].abs().max().item())
        worst = max(worst, err / tol)
        assert err < tol, (wino_form, y0, err, tol)
    print("conv_wino form %d at 1080p: worst band error %.2f of the tolerance" % (wino_form, worst))


def test_conv_wino_random_shapes(hip, wino_form):
    """30 seeded random geometries (heights 1..70, widths 4..132 in steps of 4, 33..64 input channels, 1..130 output channels,
    1..3 images, every activation the entry takes, with and without skip / clamp) against the float64 convolution"""
    rs = np.random.RandomState(7)
    hip.set_conv_wino(True)
    try:
        for it in range(30):
            n, cin, cout = int(rs.randint(1, 4)), int(rs.randint(33, 65)), int(rs.randint(1, 131))
            h, w = int(rs.randint(1, 71)), 4 * int(rs.randint(1, 34))
            act, use_res, clamp = int(rs.randint(0, 4)), bool(rs.randint(0, 2)), bool(rs.randint(0, 2))
            x, wt, b = rnd(n, cin, h, w) * float(10.0 ** rs.uniform(-3, 2)), rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5, rnd(cout)
            res = rnd(n, cout, h, w) if use_res else None
            ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
            want = (ref, F.leaky_relu(ref, 0.01), torch.relu(ref), torch.sigmoid(ref))[act]
            if act < 2:                                  # conv2d: LeakyReLU or none, skip tensor, clamp
                if use_res:
                    want = want + res.double()
                if clamp:
                    want = want.clamp(0, 1)
                wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
                out = hip.conv2d(x.cuda(), wp, bp, cout, 3, act=act, residual=res.cuda() if use_res else None, clamp=clamp)
            else:                                        # gconv2d: the generic activations
                wpg, bpg = hip.pack_gconv_weight(wt.cuda(), b.cuda())
                out = hip.gconv2d(x.cuda(), wpg, bpg, cout, 3, 1, 1, act=act)
            err = (out.double().cpu() - want).abs().max().item()
            assert err < 3e-6 * max(1.0, ref.abs().max().item()), (it, (n, cin, cout, h, w, act, use_res, clamp), err)
    finally:
        hip.set_conv_wino(False)


def test_conv_wino_graph_replay_and_second_stream(hip, wino_form):
    """the Winograd kernel allocates nothing and keeps no host state: captured in a graph and replayed, and launched on a second
    stream, it gives the eager result bit for bit"""
    n, cin, cout, h, w = 2, 64, 64, 30, 64
    x, wt, b = rnd(n, cin, h, w).cuda(), (rnd(cout, cin, 3, 3) / 24).cuda(), rnd(cout).cuda()
    wp, bp = hip.pack_conv_weight(wt, b)
    hip.set_conv_wino(True)
    try:
        want = hip.conv2d(x, wp, bp, cout, 3, act=1)
        out = torch.empty_like(want)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            hip.conv2d(x, wp, bp, cout, 3, act=1, out=out)
        torch.cuda.current_stream().wait_stream(s)
        assert torch.equal(out, want)
        out.zero_()
        with torch.cuda.graph(g):
            hip.conv2d(x, wp, bp, cout, 3, act=1, out=out)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    finally:
        hip.set_conv_wino(False)


def test_conv_ws16_row_scales(hip):
    """Per-row power-of-two scales: rows whose magnitudes differ by 10^10 inside one image, zero rows, a scale change between
    every pair of neighbouring rows; the error stays at float32-summation level RELATIVE TO EACH OUTPUT ROW's own magnitude."""
    n, cin, cout, h, w = 1, 64, 64, 36, 64
    x = rnd(n, cin, h, w)
    scale = torch.ones(h)
    scale[:6] = 1e-6
    scale[6:12] = 3e4
    scale[12:15] = 0.0
    scale[15:] = torch.tensor([2.0 ** ((i * 7) % 11 - 5) for i in range(h - 15)])      # a different exponent on every row
    x = x * scale.view(1, 1, h, 1)
    wt, b = rnd(cout, cin, 3, 3) / 24, torch.zeros(cout)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    out = hip.conv2d(x.cuda(), wp, bp, cout, 3).double().cpu()
    # an output row sums three input rows: its error scales with the largest of them
    mag = torch.stack([F.pad(scale, (1, 1))[i:i + h] for i in range(3)]).max(0).values
    err = (out - ref).abs().amax(dim=(0, 1, 3))
    assert (err <= 4e-6 * mag + 1e-30).all(), (err / (mag + 1e-30)).max().item()
    assert (out[:, :, 13] == 0).all()                       # a row whose three input rows are zero is exactly zero
    # power-of-two scaling of the input is exact: the row scales absorb it
    out2 = hip.conv2d((x * 1024.0).cuda(), wp, bp, cout, 3).double().cpu()
    assert torch.equal(out * 1024.0, out2)


def test_conv_ws16_entry_argument_checks(hip):
    import ctypes
    lib = hip.lib()
    x = torch.zeros(1, 64, 8, 32, device="cuda")
    null = ctypes.c_void_p(0)
    p = ctypes.c_void_p(x.data_ptr())
    assert lib.ct_conv3x3_ws16_f32(null, p, 0, p, null, p, 1, 64, 64, 8, 32, 0, 0, 0, 0, 0, null) == -1        # null input
    assert lib.ct_conv3x3_ws16_f32(p, p, 0, p, null, p, 1, 32, 64, 8, 32, 0, 0, 0, 0, 0, null) == -1           # cin <= 32
    assert lib.ct_conv3x3_ws16_f32(p, p, 0, p, null, p, 1, 64, 64, 8, 30, 0, 0, 0, 0, 0, null) == -3           # W % 4
    assert lib.ct_conv3x3_ws16_f32(p, p, 0, p, null, p, 0, 64, 64, 8, 32, 0, 0, 0, 0, 0, null) == 0            # empty batch


@pytest.mark.parametrize("k,c1,c2,c3", [(1, 64, 64, 1), (3, 32, 16, 5), (1, 16, 48, 33)])
def test_conv_three_sources_equals_concat(hip, k, c1, c2, c3):
    """ct_conv2d_split_f32 with three input tensors == the convolution of torch.cat([a, b, c], 1) (DCMCS3DI's
    transfer[0] reads fea_left, fea_warped and the valid mask without the 129-channel concatenation); bitwise, since the
    chunks of 16 channels and their order are the same."""
    n, h, w, cout = 2, 20, 36, 64
    a, b, c = rnd(n, c1, h, w), rnd(n, c2, h, w), rnd(n, c3, h, w)
    cin = c1 + c2 + c3
    wt, bias = rnd(cout, cin, k, k) / (cin * k * k) ** 0.5, rnd(cout)
    wp, bp = hip.pack_conv_weight(wt.cuda(), bias.cuda())
    cat = torch.cat([a, b, c], 1).cuda()
    want = hip.conv2d(cat, wp, bp, cout, k, act=1)
    got = hip.conv2d(a.cuda(), wp, bp, cout, k, act=1, x2=b.cuda(), x3=c.cuda())
    same_kernel = not (k == 3 and 32 < cin <= 64)     # a single 3x3 input with 32 < cin <= 64 takes conv_ws (other arithmetic)
    if same_kernel:
        assert torch.equal(got, want)
    ref = F.leaky_relu(F.conv2d(cat.double().cpu(), wt.double(), bias.double(), padding=k // 2), 0.01)
    assert (got.double().cpu() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    # batch views as inputs (fea[:B] of the two-view tensor)
    big = rnd(n + 1, c1, h, w).cuda()
    got = hip.conv2d(big[:n], wp, bp, cout, k, x2=b.cuda(), x3=c.cuda())
    want = hip.conv2d(torch.cat([big[:n], b.cuda(), c.cuda()], 1), wp, bp, cout, k)
    assert torch.equal(got, want) if same_kernel else (got - want).abs().max().item() < 4e-6 * max(1.0, want.abs().max().item())
    hip.set_conv_mode("exact")                       # no split kernel: the concatenation is materialised, same values to rounding
    try:
        ex = hip.conv2d(a.cuda(), wp, bp, cout, k, act=1, x2=b.cuda(), x3=c.cuda())
    finally:
        hip.set_conv_mode("split")
    assert (ex.double().cpu() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("k,cin,cout,channels,c0,act", [(1, 64, 64, 64, 0, 0), (1, 64, 64, 96, 0, 0), (1, 64, 32, 96, 64, 1),
                                                      (3, 32, 64, 64, 0, 1), (1, 129, 132, 136, 4, 0), (1, 16, 4, 8, 4, 0)])
def test_conv_rows_output_equals_transposed_nchw(hip, k, cin, cout, channels, c0, act):
    """ct_conv2d_split_rows_f32 (the q / k / v convolutions of the parallax attention store token rows from their
    epilogue) == the NCHW convolution followed by the transpose, bitwise where the same kernel computes both; the other channels
    of the rows tensor are not touched; ragged tiles, several output groups, partial 32-channel tiles, batch views."""
    n, h, w = 3, 21, 44
    x = rnd(n + 1, cin, h, w).cuda()[1:]                   # a batch view
    wt, bias = rnd(cout, cin, k, k) / (cin * k * k) ** 0.5, rnd(cout)
    wp, bp = hip.pack_conv_weight(wt.cuda(), bias.cuda())
    rows = torch.full((n * h, w, channels), 7.0, device="cuda")
    got = hip.conv2d_rows(x, wp, bp, cout, k, act=act, out=rows, c0=c0)
    assert got is rows
    want = hip.conv2d(x, wp, bp, cout, k, act=act)      # 3x3 with cin <= 32: the tile kernel here too (conv_ws starts at cin 33)
    want_rows = want.permute(0, 2, 3, 1).reshape(n * h, w, cout)
    assert torch.equal(rows[:, :, c0:c0 + cout], want_rows)
    keep = torch.ones(channels, dtype=torch.bool)
    keep[c0:c0 + cout] = False
    assert (rows[:, :, keep.cuda()] == 7.0).all()
    ref = F.conv2d(x.double().cpu(), wt.double(), bias.double(), padding=k // 2)
    ref = F.leaky_relu(ref, 0.01) if act else ref
    assert (want.double().cpu() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


def test_conv_rows_argument_checks_and_fallback(hip):
    x = rnd(1, 64, 8, 32).cuda()
    wt, bias = rnd(64, 64, 1, 1), rnd(64)
    wp, bp = hip.pack_conv_weight(wt.cuda(), bias.cuda())
    with pytest.raises(hip.CtHipError):
        hip.conv2d_rows(x, wp, bp, 64, 1, out=torch.empty(8, 32, 66, device="cuda"))           # channels % 4
    with pytest.raises(hip.CtHipError):
        hip.conv2d_rows(x, wp, bp, 64, 1, out=torch.empty(8, 32, 96, device="cuda"), c0=36)    # c0 + cout > channels
    assert hip.conv2d_rows(rnd(1, 64, 8, 30).cuda(), wp, bp, 64, 1) is None                    # W % 4: the caller transposes
    hip.set_conv_mode("exact")
    try:
        assert hip.conv2d_rows(x, wp, bp, 64, 1) is None
    finally:
        hip.set_conv_mode("split")
    lib, p, null = hip.lib(), x.data_ptr(), None
    assert lib.ct_conv2d_split_rows_f32(p, p, p, p, 1, 64, 64, 8, 32, 1, 1, 64 * 8 * 32, 66, 0, 0, 0, 0, null) == -1
    assert lib.ct_conv2d_split_rows_f32(p, p, p, p, 1, 64, 64, 8, 32, 1, 1, 64 * 8 * 32, 64, 4, 0, 0, 0, null) == -1
    assert lib.ct_conv2d_split_rows_f32(p, p, p, p, 0, 64, 64, 8, 32, 1, 1, 64 * 8 * 32, 64, 0, 0, 0, 0, null) == 0
    # cout % 4: the rows epilogue stores whole float4 groups -- a direct C-ABI caller must get CT_E_BADARG, not a write past channel cout
    assert lib.ct_conv2d_split_rows_f32(p, p, p, p, 1, 64, 62, 8, 32, 1, 1, 64 * 8 * 32, 64, 0, 0, 0, 0, null) == -1


def test_dcmcs3di_rows_path_equals_transposed_path(hip):
    """the forward pass with q / k / v written as rows by the convolutions against the one that transposes NCHW q / k / v.  Until
    round 5 both came from the same tile kernel and agreed bitwise; since round 6 the rows come from ct_conv1x1_rows_f32 (exact float32
    products, csrc/conv1x1_rows.hip) and the NCHW tensors from the two-piece tile kernel: the same values to float32 rounding, and the
    attention outputs with them (the mask away from its 0.1 threshold)."""
    from methods.dcmcs3di import DCMCS3DI
    from pasmnet.backbone import conv_forward
    torch.manual_seed(3)
    net = DCMCS3DI(extraction_layers=2, transfer_layers=1).cuda().eval()
    left, right = torch.rand(1, 3, 24, 64, device="cuda"), torch.rand(1, 3, 24, 64, device="cuda")
    parts = net.forward_parts(left, right)
    both = torch.cat([left, right], 0)
    from methods.dcmcs3di import sequential_forward
    from pasmnet.backbone import resb_forward
    fea = sequential_forward(net.extraction, both)
    head = resb_forward(net.matcher.head, fea)
    q, k = conv_forward(net.matcher.query, head), conv_forward(net.matcher.key, head)
    v = conv_forward(net.matcher.value, fea[1:])
    fw, wrgb, valid, colsum = hip.pam_streaming(q[:1].contiguous(), k[1:].contiguous(), v, right, q[1:].contiguous(), k[:1].contiguous())
    for got, want in ((parts["fea_warped"], fw), (parts["warped_rgb"], wrgb), (parts["colsum_left"], colsum)):
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    away = (colsum - 0.1).abs() > 1e-4
    assert torch.equal(parts["valid_left"][away], valid[away])


@pytest.mark.parametrize("kh,kw", [(3, 3), (1, 5), (5, 1), (1, 1)])
def test_conv_split16_wide_dynamic_range(hip, kh, kw):
    """the tile kernel in its two-piece fp16 form (csrc/conv_split.hip, F16): the running power-of-two scale of the staged
    16-channel input tiles -- channel blocks of very different magnitude in either order (the accumulators are rescaled when the
    scale drops in the middle of the contraction), all-zero blocks, image regions of different magnitude, a tiny per-layer weight
    scale; pre-activation addend, activation and skip on the unscaled value"""
    n, cin, cout, h, w = 2, 160, 96, 19, 72
    x = rnd(n, cin, h, w)
    x[:, 16:32] *= 1e-6
    x[:, 48:64] = 0.0
    x[0, 64:96] *= 3e4                    # large channels in the middle: the scale drops at chunk 4
    x[1, :16] *= 1e5                      # ... or at once
    x[:, :, :, 40:] *= 1e-3               # the right tiles live at another magnitude
    wt, b = rnd(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5 * 1e-4, rnd(cout) * 1e-3
    res, add = rnd(n, cout, h, w), rnd(n, cout, h, w) * 1e-2
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    pad = (kh // 2, kw // 2)
    lin = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=pad) + 1e-300       # sum |x||w|
    # the staged tile is 16 channels x (8 + kh - 1) rows x (32 + 8) columns: a value keeps 2^-36 of its tile's maximum
    floor = 2.0 ** -34 * F.max_pool2d(x.abs().amax(dim=1, keepdim=True), (2 * 12 + 1, 2 * 44 + 1), 1, (12, 44)).double() * wt.abs().sum().item() / cout
    got = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, pad).double().cpu()
    assert ((got - lin).abs() / (1e-6 * bound + floor + 1e-7 * b.abs().max().item())).max().item() < 1.0
    got = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, pad, act=3, addend=add.cuda()).double().cpu()      # sigmoid(conv + addend)
    assert (got - torch.sigmoid(lin + add.double())).abs().max().item() < 2e-6
    got = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, pad, residual=res.cuda()).double().cpu()
    assert ((got - (lin + res.double())).abs() / (1e-6 * bound + floor + 3e-7 * (1 + res.abs().double()))).max().item() < 1.0
    hip.set_conv_ws16(False)              # the three-piece bf16 form computes the same layer
    try:
        old = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, pad).double().cpu()
    finally:
        hip.set_conv_ws16(True)
    assert ((old - lin).abs() / (1e-6 * bound + 1e-7 * b.abs().max().item())).max().item() < 1.0


def test_conv_post_ops_equal_the_elementwise_kernels(hip):
    """the GRU's elementwise steps in the convolution epilogue (post_op of ct_conv2d_split_f32: sigmoid(.) * h and
    (1 - z) h + z tanh(.)) == the convolution followed by the elementwise kernel, bitwise (same float32 expressions)"""
    n, cin, cout, h, w = 2, 128, 128, 24, 40
    x, hid, z, add = rnd(n, cin, h, w), rnd(n, cout, h, w), torch.rand(n, cout, h, w, generator=G), rnd(n, cout, h, w)
    wt, b = rnd(cout, cin, 1, 5) / (cin * 5) ** 0.5, rnd(cout)
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    xc, hc, zc, ac = x.cuda(), hid.cuda(), z.cuda(), add.cuda()
    r = hip.gconv2d(xc, wp, bp, cout, (1, 5), 1, (0, 2), act=3, addend=ac)
    rh = hip.gconv2d(xc, wp, bp, cout, (1, 5), 1, (0, 2), act=3, addend=ac, post=(1, hc, None))
    assert torch.equal(rh, hip.eltwise(1, r, hc))
    q = hip.gconv2d(xc, wp, bp, cout, (1, 5), 1, (0, 2), act=4, addend=ac)
    hn = hip.gconv2d(xc, wp, bp, cout, (1, 5), 1, (0, 2), act=4, addend=ac, post=(2, zc, hc))
    assert torch.equal(hn, hip.eltwise(2, zc, hc, q))
    ref = (1 - z.double()) * hid.double() + z.double() * torch.tanh(F.conv2d(x.double(), wt.double(), b.double(), padding=(0, 2)) + add.double())
    assert (hn.double().cpu() - ref).abs().max().item() < 3e-6
    with pytest.raises(hip.CtHipError):
        hip.gconv2d(xc, wp, bp, cout, (1, 5), 1, (0, 2), act=4, addend=ac, post=(2, zc, hc[:, :64]))


SK_CASES = [   # (cin, cout, kh, kw, h, w): (tile, 64-channel) units / 512 resident workgroups -> rounds the whole-unit form would take
    (256, 128, 1, 5, 136, 240),    # 544 units: 1.06 rounds (the SepConvGRU at a 544 x 960 inference size)
    (160, 128, 5, 1, 136, 240),    # ... with 10 chunks per unit: shares of 10.6 stages
    (128, 96, 3, 3, 136, 240),     # 544 units, the second group half empty
    (64, 192, 3, 3, 136, 240),     # 816 units, three groups per tile
    (48, 64, 1, 1, 200, 328),      # 550 units of three stages each
    (32, 64, 3, 3, 264, 448),      # 924 units of two stages: a share is 3.6 stages (whole units between the two pieces)
]


@pytest.mark.parametrize("cfg", SK_CASES)
def test_conv_split_stream_k(hip, cfg):
    """launches whose units do not fill whole rounds of the resident workgroups share the units' channel loops between neighbouring
    workgroups (csrc/conv_split.hip, stream-K): the whole-unit form's result within the kernel's bound (the pieces of a shared unit carry their own fp16
    scales and meet in one float32 addition), against float64 within the same bound, repeatable bit for bit, flag words zero afterwards -- plain, with a second
    input tensor, with a pre-activation addend + activation + GRU post-op, and with a skip tensor"""
    cin, cout, kh, kw, h, w = cfg
    n = 2
    x, wt, b = rnd(n, cin, h, w), rnd(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5, rnd(cout)
    x[:, : cin // 2] *= 50.0                 # the head piece of a split unit lives at another scale than its tail
    add, hid, res = rnd(n, cout, h, w), rnd(n, cout, h, w), rnd(n, cout, h, w)
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    pad = (kh // 2, kw // 2)
    xc, ac, hc, rc = x.cuda(), add.cuda(), hid.cuda(), res.cuda()
    lin = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=pad)

    def run():
        outs = [hip.gconv2d(xc, wp, bp, cout, (kh, kw), 1, pad)]
        if cin % 32 == 0:
            outs.append(hip.gconv2d(xc[:, : cin // 2].contiguous(), wp, bp, cout, (kh, kw), 1, pad, x2=xc[:, cin // 2:].contiguous()))
        outs.append(hip.gconv2d(xc, wp, bp, cout, (kh, kw), 1, pad, act=3, addend=ac, post=(1, hc, None)))
        outs.append(hip.gconv2d(xc, wp, bp, cout, (kh, kw), 1, pad, residual=rc))
        return outs
    hip.set_conv_stream_k(False)
    try:
        whole = run()
    finally:
        hip.set_conv_stream_k(True)
    shared, again = run(), run()
    assert hip.conv_stream_k_state() == (0, 0)
    refs = [lin] + ([lin] if cin % 32 == 0 else []) + [torch.sigmoid(lin + add.double()) * hid.double(), lin + res.double()]
    assert any(not torch.equal(a_, b_) for a_, b_ in zip(shared, whole))          # the launch did take the shared form
    for got, rep, ref_whole, ref in zip(shared, again, whole, refs):
        assert torch.equal(got, rep)
        # both forms are within the kernel's bound of the truth; the tail piece of a shared unit even runs at its own (finer) scale
        assert ((got.double() - ref_whole.double()).cpu().abs() / (1.5e-6 * bound + 2e-6)).max().item() < 1.0
        assert ((got.double().cpu() - ref).abs() / (1.5e-6 * bound + 2e-6)).max().item() < 1.0


def test_conv_split_stream_k_many_launches_one_scratch(hip):
    """the scratch is owned by the stream: launches of different shapes back to back (each leaves the flag words zero), and a
    second stream gets a scratch of its own"""
    shapes = [(128, 128, 1, 5, 136, 240), (64, 192, 3, 3, 136, 240), (128, 128, 5, 1, 136, 240)]
    data = []
    for cin, cout, kh, kw, h, w in shapes:
        x, wt, b = rnd(2, cin, h, w).cuda(), (rnd(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5).cuda(), rnd(cout).cuda()
        data.append((x, hip.pack_gconv_weight(wt, b), cout, (kh, kw), (kh // 2, kw // 2)))
    first = [hip.gconv2d(x, p[0], p[1], co, k, 1, pd) for x, p, co, k, pd in data]
    for _ in range(3):
        for (x, p, co, k, pd), ref in zip(data, first):
            assert torch.equal(hip.gconv2d(x, p[0], p[1], co, k, 1, pd), ref)
    assert hip.conv_stream_k_state() == (0, 0)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        other = [hip.gconv2d(x, p[0], p[1], co, k, 1, pd) for x, p, co, k, pd in data]
        side.synchronize()
        assert hip.conv_stream_k_state() == (0, 0)
    side.synchronize()
    for a_, b_ in zip(other, first):
        assert torch.equal(a_, b_)
    assert len(hip._sk_cache) >= 2


def test_conv_split_stream_k_graph_replay_and_busy_gpu(hip):
    """the shared form is plain stream work whose scratch cleans itself: captured in a hipGraph and replayed on new data it gives the
    eager result bit for bit; with another stream keeping the CUs busy the producers are only delayed (workgroups are dispatched in
    index order: a consumer's producer is never behind it) -- same bits, no give-up, flag words zero"""
    cin, cout, h, w = 128, 128, 136, 240
    x = rnd(2, cin, h, w).cuda()
    wt, b = (rnd(cout, cin, 1, 5) / (cin * 5) ** 0.5).cuda(), rnd(cout).cuda()
    wp, bp = hip.pack_gconv_weight(wt, b)
    out = torch.empty((2, cout, h, w), device="cuda")
    hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2), out=out)              # warm
    torch.cuda.synchronize()
    # the scratch of the CAPTURE stream is made before the capture (one allocated inside a capture would live in that
    # graph's private pool and be shared with later graphs through the binding's cache); without it a captured launch runs
    # without stream-K
    cap = torch.cuda.Stream()
    n_before = len(hip._sk_cache)
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=cap):
        hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2), out=out)
    assert len(hip._sk_cache) == n_before                                  # nothing was cached from inside the capture
    g0.replay()
    torch.cuda.synchronize()
    whole = hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2))
    assert (out - whole).abs().max().item() <= 2e-5                        # whole units instead of shared ones: same bound, other bits
    del g0
    assert hip.conv_scratch_prepare(stream=cap) is not None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2), out=out)
    for _ in range(3):
        x2 = rnd(2, cin, h, w).cuda()
        want = hip.gconv2d(x2, wp, bp, cout, (1, 5), 1, (0, 2)).clone()
        x.copy_(x2)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    want = hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2)).clone()
    a = torch.randn(8192, 8192, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):
            a = a @ a * 1e-4
    for _ in range(8):
        assert torch.equal(hip.gconv2d(x, wp, bp, cout, (1, 5), 1, (0, 2)), want)
    torch.cuda.synchronize()
    assert hip.conv_stream_k_state() == (0, 0)


@pytest.mark.parametrize("cfg", [(2, 64, 9, 64, 64, 0, 0), (1, 64, 5, 100, 96, 32, 1), (3, 32, 4, 36, 32, 0, 0), (1, 4, 3, 8, 8, 4, 2),
                                 (2, 60, 7, 1920, 64, 0, 0), (1, 64, 1, 4, 128, 64, 3)])
def test_conv1x1_rows_vs_float64(hip, cfg):
    """ct_conv1x1_rows_f32 (csrc/conv1x1_rows.hip: the query / key / value 1x1 convolutions in front of the parallax attention as a
    streaming kernel on the exact-f32 matrix pipe; reference pasmnet/attention.py:39-40,44-45, dcmcs3di.py:58) against the float64
    convolution: widths that are not multiples of the 32-pixel segment, cout below 32 / 64, rows wider than cout with an offset (the
    value rows carry RGB behind the 64 channels), every activation, several images; nothing outside the written channels changes."""
    n, cout, h, w, channels, c0, act = cfg
    x, wt, b = rnd(n, 64, h, w) * 2, rnd(cout, 64, 1, 1) / 8, rnd(cout)
    ref = F.conv2d(x.double(), wt.double(), b.double())
    want = (ref, F.leaky_relu(ref, 0.01), torch.relu(ref), torch.sigmoid(ref))[act].permute(0, 2, 3, 1).reshape(n * h, w, cout)
    wp, bp = hip.pack_conv_weight(wt.cuda(), b.cuda())
    out = torch.full((n * h, w, channels), 777.0, device="cuda")
    got = hip.conv2d_rows(x.cuda(), wp, bp, cout, 1, act=act, out=out, c0=c0, channels=channels, raw=(wt.cuda(), b.cuda()))
    assert got is out
    torch.cuda.synchronize()
    o = out.cpu().double()
    err = (o[:, :, c0:c0 + cout] - want).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.abs().max().item()), (cfg, err)
    assert bool((o[:, :, :c0] == 777.0).all()) and bool((o[:, :, c0 + cout:] == 777.0).all())
    # a view of a larger batch (the two views of a stereo pair live in one tensor) and no bias
    big = rnd(n + 1, 64, h, w).cuda()
    got2 = hip.conv2d_rows(big[1:], wp, bp, cout, 1, out=None, raw=(wt.cuda(), None))
    want2 = F.conv2d(big[1:].cpu().double(), wt.double()).permute(0, 2, 3, 1).reshape(n * h, w, cout)
    assert (got2.cpu().double() - want2).abs().max().item() < 2e-6 * max(1.0, want2.abs().max().item())
