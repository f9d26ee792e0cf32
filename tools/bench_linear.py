#!/usr/bin/env python3
"""Time nn.Linear on tokens: float32 MFMA ("exact"), three-piece bf16 LDS-tiled ("split"), resident-weight two-piece fp16 where the
shape has it ("ws16"), on the GMFlow shapes at 960x540."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip

def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

torch.manual_seed(0)
MODES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["exact", "split", "ws16"]
ROT = 6                                          # distinct input buffers, cycled: every call streams its tokens from HBM, as in the network
SHAPES = [(T, k, n, act) for T in (14336, 114688) for k, n, act in ((128, 128, 0), (128, 384, 0), (256, 1024, 6), (256, 1024, 0), (1024, 128, 0))]
if len(sys.argv) > 2:                            # one shape: T,K,N,act
    SHAPES = [tuple(int(v) for v in sys.argv[2].split(","))]
for T, k, n, act in SHAPES:
    if True:
        xs = [torch.randn(T, k, device="cuda") for _ in range(ROT)]
        x = xs[0]; w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        if act == 6: ref = torch.nn.functional.gelu(ref)
        line = "T=%6d K=%4d N=%4d" % (T, k, n)
        for mode in MODES:
            ct_hip.set_linear_ws16(mode == "ws16")
            kw = dict(mode="split", partials=True) if mode == "ws16" else dict(mode=mode)
            out = ct_hip.linear_tokens(x, w, b, act=act, **kw)
            err = ((out.sum(0) if out.dim() == 3 else out).double() - ref).abs().max().item()
            del out
            it = [0]
            def call():
                it[0] += 1
                return ct_hip.linear_tokens(xs[it[0] % ROT], w, b, act=act, **kw)
            ms = t_ms(call, 24)
            line += "  %s %.3f ms %6.1f TFLOP/s err %.2g" % (mode, ms, 2.0 * T * k * n / ms / 1e9, err)
        print(line)

# the q / k / v projections: three separate launches vs one three-slice launch
for T in (14336, 114688):
    xs = [torch.randn(T, 128, device="cuda") for _ in range(ROT)]
    ws = [torch.randn(128, 128, device="cuda") / 128 ** 0.5 for _ in range(3)]
    it = [0]
    def sep():
        it[0] += 1
        return [ct_hip.linear_tokens(xs[it[0] % ROT], w, None, mode="split") for w in ws]
    def multi():
        it[0] += 1
        return ct_hip.linear_tokens_multi(xs[it[0] % ROT], ws)
    def one():
        it[0] += 1
        return ct_hip.linear_tokens_multi(xs[it[0] % ROT], ws[:1])
    ct_hip.set_linear_ws16(True)
    print("T=%6d q/k/v 128->128: separate %.3f ms  one launch %.3f ms;  single layer: wres %.3f ms  ws16 %.3f ms" % (
        T, t_ms(sep, 24), t_ms(multi, 24), t_ms(lambda: (it.__setitem__(0, it[0] + 1), ct_hip.linear_tokens(xs[it[0] % ROT], ws[0], None, mode="split"))[1], 24), t_ms(one, 24)))
