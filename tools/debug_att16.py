import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "color-transfer_amd")):
    sys.path.insert(0, p)
import torch
import ct_hip
G = torch.Generator().manual_seed(5)
def rnd(*s): return torch.randn(*s, generator=G)
def run(name, q, k, v):
    scores = torch.matmul(q.double(), k.double().transpose(1, 2)) / 128 ** 0.5
    p = torch.softmax(scores, dim=-1)
    ref = torch.matmul(p, v.double())
    mag = torch.matmul(p, v.double().abs()).clamp_min(1e-300)
    out = ct_hip.attention_tokens(q.cuda(), k.cuda(), v.cuda(), None, nsplit=1).double().cpu()
    e = ((out - ref).abs() / mag)
    i = e.argmax().item()
    print("%-28s max rel err %.3e at flat %d (token %d, ch %d) out %.4e ref %.4e  nan=%d" % (name, e.max().item(), i, (i // v.shape[2]) % q.shape[1], i % v.shape[2], out.flatten()[i], ref.flatten()[i], torch.isnan(out).sum().item()))
b, l = 1, 416
q, k, v = rnd(b, l, 128), rnd(b, l, 128), rnd(b, l, 128)
run("plain", q, k, v)
run("q tiny k large", q * 3e-5, k * 2e4, v)
q2 = q.clone(); q2[0, :100] *= 40
run("q rows x40", q2, k, v)
k2 = k.clone(); k2[0, 200:232] *= 1e-3
run("k tile small", q, k2, v)
v2 = v.clone(); v2[0, :64] = 0
run("v zero tiles first", q, k, v2)
v3 = v.clone(); v3[0, 64:128] *= 1e-12
run("v tiny tiles", q, k, v3)
v4 = v.clone(); v4[0, 300:] *= 1e9
run("v huge late", q, k, v4)
run("v logspace up", q, k, v * torch.logspace(-6, 6, l)[:, None])
run("v logspace down", q, k, v * torch.logspace(6, -6, l)[:, None])
run("v x1e-20", q, k, v * 1e-20)
run("v x1e20", q, k, v * 1e20)
