"""Regrain (the second half of automated_color_grading), numpy float64 restatement of the reference's
methods/iterative.py:62-138.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

`resize` restates scikit-image 0.18.3's skimage.transform.resize for the one way the reference calls it
(`resize(arr, (h2, w2))` on an H x W x 3 float array: order 1, mode 'reflect', anti_aliasing on when shrinking): third-party
code (unpinned in requirements.txt), pinned here by tests/golden/make_golden_regrain.py which runs the real skimage and the
reference's own `_regrain` / `_solve` / `automated_color_grading`.  scipy.ndimage.gaussian_filter is called exactly as
skimage calls it (_warps.py: `ndi.gaussian_filter(image, sigma, cval=0, mode='mirror')`).
"""
import numpy as np
import scipy.ndimage as ndi


def _reflect(coord, dim):                      # skimage _warp_fast coord_map(dim, coord, 'R')
    cmax = dim - 1
    c = np.asarray(coord).copy()
    if dim == 1:
        return np.zeros_like(c)
    neg = c < 0
    n = -c[neg]
    c[neg] = np.where((n // cmax) % 2 != 0, cmax - (n % cmax), n % cmax)
    big = c > cmax
    b = c[big]
    c[big] = np.where((b // cmax) % 2 != 0, cmax - (b % cmax), b % cmax)
    return c


def resize(image, output_shape):
    image = np.asarray(image, dtype=np.float64)
    hi, wi = image.shape[:2]
    ho, wo = output_shape
    fr, fc = hi / ho, wi / wo
    sigma = (max(0.0, (fr - 1) / 2), max(0.0, (fc - 1) / 2), 0.0)
    image = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    sr, sc = fr * (np.arange(ho) + 0.5) - 0.5, fc * (np.arange(wo) + 0.5) - 0.5
    minr, minc = np.floor(sr).astype(int), np.floor(sc).astype(int)
    maxr, maxc = np.ceil(sr).astype(int), np.ceil(sc).astype(int)
    dr, dc = (sr - minr)[:, None, None], (sc - minc)[None, :, None]
    r0, r1, c0, c1 = _reflect(minr, hi), _reflect(maxr, hi), _reflect(minc, wi), _reflect(maxc, wi)
    top = (1 - dc) * image[r0][:, c0] + dc * image[r0][:, c1]
    bottom = (1 - dc) * image[r1][:, c0] + dc * image[r1][:, c1]
    return (1 - dr) * top + dr * bottom


def solve(out, img_in, col, nbit, level, eps=1e-6):                       # iterative.py:81-117
    img_in, col, out = (np.asarray(a, dtype=np.float64) for a in (img_in, col, out))
    c = img_in.shape[2]
    first0 = lambda a: np.concatenate((a[:1, :], a[:-1, :]), axis=0)     # noqa: E731
    first1 = lambda a: np.concatenate((a[:, :1], a[:, :-1]), axis=1)     # noqa: E731
    last0 = lambda a: np.concatenate((a[1:, :], a[-1:, :]), axis=0)      # noqa: E731
    last1 = lambda a: np.concatenate((a[:, 1:], a[:, -1:]), axis=1)      # noqa: E731
    dx, dy = last1(img_in) - first1(img_in), last0(img_in) - first0(img_in)
    delta = np.sqrt((dx ** 2 + dy ** 2).sum(axis=2, keepdims=True))
    psi = 256 * delta / 5
    psi[psi > 1] = 1
    phi = 30 * 2 ** (-level) / (1 + 10 * delta)
    phi1, phi2, phi3, phi4 = (last1(phi) + phi) / 2, (last0(phi) + phi) / 2, (first1(phi) + phi) / 2, (first0(phi) + phi) / 2
    rho = 1 / 5.0
    for _ in range(nbit):
        den = psi + phi1 + phi2 + phi3 + phi4
        num = (psi * col + phi1 * (last1(out) - last1(img_in) + img_in) + phi2 * (last0(out) - last0(img_in) + img_in)
               + phi3 * (first1(out) - first1(img_in) + img_in) + phi4 * (first0(out) - first0(img_in) + img_in))
        out = num / np.tile(den + eps, [1, 1, c]) * (1 - rho) + rho * out
    return out


def regrain(img_in, col, nbits=(4, 16, 32, 64, 64, 64), level=0):         # iterative.py:62-78
    h, w, _ = img_in.shape
    h2, w2 = (h + 1) // 2, (w + 1) // 2
    if len(nbits) > 1 and h2 > 20 and w2 > 20:
        out = resize(regrain(resize(img_in, (h2, w2)), resize(col, (h2, w2)), nbits[1:], level + 1), (h, w))
    else:
        out = img_in
    return solve(out, img_in, col, nbits[0], level)
