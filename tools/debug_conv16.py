import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import ct_hip as hip
G = torch.Generator().manual_seed(11)
def rnd(*s): return torch.randn(*s, generator=G)
def run(name, x, wt, b, kh=3, kw=3):
    cout = wt.shape[0]
    wp, bp = hip.pack_gconv_weight(wt.cuda(), b.cuda())
    pad = (kh // 2, kw // 2)
    lin = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=pad) + 1e-300
    for f16 in (True, False):
        hip.set_conv_ws16(f16)
        got = hip.gconv2d(x.cuda(), wp, bp, cout, (kh, kw), 1, pad).double().cpu()
        e = (got - lin).abs() / bound
        i = e.argmax().item()
        idx = torch.unravel_index(torch.tensor(i), e.shape)
        print("%-30s f16=%d max err/bound %.3e at %s got %.5e ref %.5e" % (name, f16, e.max().item(), [int(t) for t in idx], got.flatten()[i], lin.flatten()[i]))
    hip.set_conv_ws16(True)
n, cin, cout, h, w = 2, 160, 96, 19, 72
x = rnd(n, cin, h, w); wt = rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5; b = torch.zeros(cout)
run("plain", x, wt, b)
run("w x1e-4", x, wt * 1e-4, b)
x1 = x.clone(); x1[:, 16:32] *= 1e-6; run("block tiny", x1, wt, b)
x2 = x.clone(); x2[:, 48:64] = 0; run("block zero", x2, wt, b)
x3 = x.clone(); x3[0, 64:96] *= 3e4; run("blocks large mid", x3, wt, b)
x4 = x.clone(); x4[1, :16] *= 1e5; run("block large first", x4, wt, b)
x5 = x.clone(); x5[:, :, :, 40:] *= 1e-3; run("right tiles small", x5, wt, b)
run("bias", x, wt, rnd(cout))
print("combined")
x = rnd(n, cin, h, w)
x[:, 16:32] *= 1e-6
x[:, 48:64] = 0.0
x[0, 64:96] *= 3e4
x[1, :16] *= 1e5
x[:, :, :, 40:] *= 1e-3
wt, b = rnd(cout, cin, 3, 3) / (cin * 9) ** 0.5 * 1e-4, rnd(cout) * 1e-3
run("combined", x, wt, b)
run("combined nobias", x, wt, b * 0)
xx = x.clone(); xx[:, :, :, 40:] *= 1e3
run("combined, uniform columns", xx, wt, b)
