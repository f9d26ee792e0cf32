"""Error of Winograd F(2x2,3x3) on two fp16 pieces per float32 operand (hi*hi + hi*lo + lo*hi, float32 accumulation), against the
float64 direct convolution; the direct two-piece form beside it.  64 -> 64 channels, random data with rows of mixed magnitude."""
import numpy as np
rng = np.random.default_rng(0)
C, K, H, W = 64, 64, 36, 36
x = rng.standard_normal((C, H, W)).astype(np.float32) * 3
x[:, :H // 2] *= 1e-2
w = (rng.standard_normal((K, C, 3, 3)) / 24).astype(np.float32)

def split16(a, scale):
    s = a * scale
    hi = s.astype(np.float16)
    lo = (s - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

def mm3(ah, al, bh, bl):      # three products, float32 accumulate (einsum over channel dim c)
    f = lambda p, q: np.einsum("kc,cn->kn", p, q, dtype=np.float32)
    return f(al, bh) + f(ah, bl) + f(ah, bh)

# float64 reference (valid conv, no padding: interior only)
xd, wd = x.astype(np.float64), w.astype(np.float64)
ref = np.zeros((K, H - 2, W - 2))
for dy in range(3):
    for dx in range(3):
        ref += np.einsum("kc,chw->khw", wd[:, :, dy, dx], xd[:, dy:dy + H - 2, dx:dx + W - 2])

# direct two-piece: per-row scale of the input (max -> [2^11, 2^12)), per-layer weight scale
def pow2_scale(m): return 2.0 ** (11 - np.floor(np.log2(m)))
ws = pow2_scale(np.abs(w).max())
wh, wl = split16(w, ws)
direct = np.zeros((K, H - 2, W - 2), np.float32)
for y in range(H - 2):
    acc = np.zeros((K, W - 2), np.float32)
    for dy in range(3):
        row = x[:, y + dy]
        rs = pow2_scale(max(np.abs(row).max(), 1e-30))
        rh, rl = split16(row, rs)
        part = np.zeros((K, W - 2), np.float32)
        for dx in range(3):
            part += mm3(wh[:, :, dy, dx], wl[:, :, dy, dx], rh[:, dx:dx + W - 2], rl[:, dx:dx + W - 2])
        acc += part / np.float32(rs * ws)
    direct[:, y] = acc

# Winograd F(2x2, 3x3)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
U = np.einsum("ij,kcjl,ml->kcim", G, wd, G)                    # [K, C, 4, 4] in float64, rounded once to float32 (host side)
U = U.astype(np.float32)
us = np.array([[pow2_scale(np.abs(U[:, :, i, j]).max()) for j in range(4)] for i in range(4)])     # one scale per transform position
Uh = np.empty_like(U); Ul = np.empty_like(U)
for i in range(4):
    for j in range(4):
        Uh[:, :, i, j], Ul[:, :, i, j] = split16(U[:, :, i, j], us[i, j])
ty, tx = (H - 2) // 2, (W - 2) // 2
wino = np.zeros((K, H - 2, W - 2), np.float32)
for by in range(ty):
    # a strip of tiles (one tile row): d[c, tile, 4, 4]
    d = np.stack([x[:, 2 * by:2 * by + 4, 2 * bx:2 * bx + 4] for bx in range(tx)], axis=1)       # [C, tx, 4, 4]
    V = np.einsum("ij,ctjl,ml->ctim", Bt, d, Bt).astype(np.float32)                              # float32 adds
    M = np.zeros((K, tx, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            vs = pow2_scale(max(np.abs(V[:, :, i, j]).max(), 1e-30))                               # scale per position and tile row
            vh, vl = split16(V[:, :, i, j], vs)
            M[:, :, i, j] = mm3(Uh[:, :, i, j], Ul[:, :, i, j], vh, vl) / np.float32(vs * us[i, j])
    Y = np.einsum("ij,ktjl,ml->ktim", At, M, At).astype(np.float32)                              # [K, tx, 2, 2]
    for bx in range(tx):
        wino[:, 2 * by:2 * by + 2, 2 * bx:2 * bx + 2] = Y[:, bx]
rng_out = np.abs(ref).max()
for name, got in (("direct two-piece", direct), ("winograd two-piece", wino), ("float32 direct (numpy)", None)):
    if got is None:
        got = np.zeros((K, H - 2, W - 2), np.float32)
        for dy in range(3):
            for dx in range(3):
                got += np.einsum("kc,chw->khw", w[:, :, dy, dx], x[:, dy:dy + H - 2, dx:dx + W - 2], dtype=np.float32)
    e = np.abs(got - ref)
    small = e[:, :H // 2 - 3].max() / np.abs(ref[:, :H // 2 - 3]).max()
    print("%-24s max err / output range %.2e   (small-magnitude rows: %.2e of their own range)" % (name, e.max() / rng_out, small))
