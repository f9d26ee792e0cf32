"""Per-frame quality metrics of the reference's Runner.test_step (methods/__init__.py:29-40), float64 CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* `psnr`  piq.psnr(x, y): data_range 1, mean squared error over C,H,W per sample, 10 log10(1 / mse).
* `ssim`  piq.ssim(x, y) with piq's defaults.  piq is a third-party dependency (unpinned, requirements.txt) that is absent
          offline: restated from its published source (piq/ssim.py: `ssim`, `_ssim_per_channel`; piq/functional/filters.py:
          `gaussian_filter`).  The core (Gaussian-weighted Wang SSIM on "valid" windows) is ANCHORED on scikit-image 0.18.3's
          `structural_similarity(gaussian_weights=True, sigma=1.5, use_sample_covariance=False)` by
          tests/golden/make_golden_ssim.py; piq's average-pool downsampling in front of it is restated only (parity unpinned).
* `icid`  utils/icid.py:28-152 line by line; its two third-party calls are restated here (parity unpinned for them):
          kornia.color.rgb_to_lab (kornia/color/lab.py + rgb.py: sRGB 0.04045 / 2.4, XYZ matrix, D65 white, 0.008856 / 7.787 /
          4/29) and torchvision.transforms.functional.gaussian_blur (kernel1d = normalised exp(-0.5 (x / sigma)^2) on
          linspace(-(k-1)/2, (k-1)/2, k), reflect padding, depth-wise conv).  tests/golden/make_golden_icid.py RUNS the
          reference's utils/icid.py with exactly these two restatements plugged in, so every line of icid.py itself is pinned.
* `fsim`  piq.fsim(x, y) with piq's defaults (chromatic FSIMc, 4 scales x 4 orientations of log-Gabor filters, Kovesi's
          phase congruency PC_2 with the median noise estimate, Scharr gradient magnitude), restated from piq/fsim.py
          (`fsim`, `_construct_filters`, `_phase_congruency`, `_lowpassfilter`) and piq/functional (`get_meshgrid`,
          `ifftshift`, `scharr_filter`, `gradient_map`, `similarity_map`, `rgb2yiq`).  piq is absent offline and there is
          nothing in this image to anchor it on: PARITY UNPINNED.
"""
import math

import torch
import torch.nn.functional as F


def psnr(x, y):
    mse = ((x.double() - y.double()) ** 2).flatten(1).mean(dim=1)
    return 10.0 * torch.log10(1.0 / mse)


def metric_factor(h, w):
    return max(1, round(min(h, w) / 256))


def _gaussian_2d(size, sigma, dtype):
    c = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = c ** 2
    g = torch.exp(-(g.unsqueeze(0) + g.unsqueeze(1)) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(x, y, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03, downsample=True):
    """piq.ssim(x, y, data_range=1., reduction='none') -> [B]"""
    x, y = x.double(), y.double()
    f = metric_factor(*x.shape[-2:])
    if f > 1 and downsample:
        x, y = F.avg_pool2d(x, kernel_size=f), F.avg_pool2d(y, kernel_size=f)
    c = x.shape[1]
    kernel = _gaussian_2d(kernel_size, kernel_sigma, x.dtype).repeat(c, 1, 1, 1)
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = F.conv2d(x, kernel, groups=c), F.conv2d(y, kernel, groups=c)
    mu_xx, mu_yy, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    sigma_xx = F.conv2d(x ** 2, kernel, groups=c) - mu_xx
    sigma_yy = F.conv2d(y ** 2, kernel, groups=c) - mu_yy
    sigma_xy = F.conv2d(x * y, kernel, groups=c) - mu_xy
    cs = (2.0 * sigma_xy + c2) / (sigma_xx + sigma_yy + c2)
    ss = (2.0 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.mean(dim=(-1, -2)).mean(dim=1)


def rgb_to_lab(image):                                             # kornia.color.rgb_to_lab
    lin = torch.where(image > 0.04045, torch.pow((image + 0.055) / 1.055, 2.4), image / 12.92)
    r, g, b = lin[..., 0, :, :], lin[..., 1, :, :], lin[..., 2, :, :]
    xyz = torch.stack([0.412453 * r + 0.357580 * g + 0.180423 * b, 0.212671 * r + 0.715160 * g + 0.072169 * b,
                       0.019334 * r + 0.119193 * g + 0.950227 * b], dim=-3)
    white = torch.tensor([0.95047, 1.0, 1.08883], dtype=image.dtype)[..., :, None, None]
    n = xyz / white
    t = 0.008856
    v = torch.where(n > t, torch.pow(n.clamp(min=t), 1 / 3.0), 7.787 * n + 4.0 / 29.0)
    x, y, z = v[..., 0, :, :], v[..., 1, :, :], v[..., 2, :, :]
    return torch.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], dim=-3)


def gaussian_blur(img, kernel_size, sigma):                        # torchvision.transforms.functional.gaussian_blur
    def k1d(k, s):
        half = (k - 1) * 0.5
        x = torch.linspace(-half, half, steps=k, dtype=img.dtype)
        pdf = torch.exp(-0.5 * (x / s) ** 2)
        return pdf / pdf.sum()
    kx, ky = k1d(kernel_size[0], sigma[0]), k1d(kernel_size[1], sigma[1])
    kernel = torch.mm(ky[:, None], kx[None, :])
    shape = img.shape
    x = img.reshape(-1, 1, shape[-2], shape[-1]) if img.dim() != 4 else img.reshape(-1, 1, shape[-2], shape[-1])
    pad = [kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2]
    x = F.conv2d(F.pad(x, pad, mode="reflect"), kernel[None, None])
    return x.reshape(shape)


def icid(img1, img2):
    """utils/icid.py:28-152 with intent="perceptual", omit_maps67=False, downsampling=True -> scalar (mean over the batch and
    positions, like the reference); use a batch of one for a per-frame value"""
    img1, img2 = img1.double(), img2.double()
    # the reference builds its weights with torch.tensor([...]) -> float32 (0.002 becomes 0.00200000009499...), whatever
    # the dtype of the images (utils/icid.py:44)
    w = [float(v) for v in torch.tensor([0.002, 10, 10, 0.002, 0.002, 10, 10], dtype=torch.float32)]
    h, wd = img1.shape[-2:]
    f = metric_factor(h, wd)
    if f > 1:
        img1 = F.interpolate(img1, scale_factor=1 / f, mode="bilinear")
        img2 = F.interpolate(img2, scale_factor=1 / f, mode="bilinear")
    img1, img2 = rgb_to_lab(img1), rgb_to_lab(img2)
    L1, A1, B1 = img1[..., 0, :, :], img1[..., 1, :, :], img1[..., 2, :, :]
    L2, A2, B2 = img2[..., 0, :, :], img2[..., 1, :, :], img2[..., 2, :, :]
    C1, C2 = torch.sqrt(A1 ** 2 + B1 ** 2), torch.sqrt(A2 ** 2 + B2 ** 2)
    ks, sg = [11, 11], [2.0, 2.0]
    blur = lambda t: gaussian_blur(t, ks, sg)     # noqa: E731
    muL1, muC1, muL2, muC2 = blur(L1), blur(C1), blur(L2), blur(C2)
    sL1q = (blur(L1 ** 2) - muL1 ** 2).clamp(min=0)
    sL2q = (blur(L2 ** 2) - muL2 ** 2).clamp(min=0)
    sC1q = (blur(C1 ** 2) - muC1 ** 2).clamp(min=0)
    sC2q = (blur(C2 ** 2) - muC2 ** 2).clamp(min=0)
    sL1, sL2, sC1, sC2 = sL1q.sqrt(), sL2q.sqrt(), sC1q.sqrt(), sC2q.sqrt()
    dLq, dCq = (muL1 - muL2) ** 2, (muC1 - muC2) ** 2
    H = ((A1 - A2) ** 2 + (B1 - B2) ** 2 - (C1 - C2) ** 2).clamp(min=0)
    dHq = blur(torch.sqrt(H)) ** 2
    sL12 = blur(L1 * L2) - muL1 * muL2
    sC12 = blur(C1 * C2) - muC1 * muC2
    maps = [1 / (w[0] * dLq + 1), (w[1] + 2 * sL1 * sL2) / (w[1] + sL1q + sL2q), ((w[2] + sL12.abs()) / (w[2] + sL1 * sL2)) ** 3,
            1 / (w[3] * dCq + 1), 1 / (w[4] * dHq + 1), (w[5] + 2 * sC1 * sC2) / (w[5] + sC1 ** 2 + sC2 ** 2),
            (w[6] + sC12.abs()) / (w[6] + sC1 * sC2)]
    prod = maps[0]
    for m in maps[1:]:
        prod = prod * m
    return 1 - prod.mean()



# ---- FSIM (piq/fsim.py) -------------------------------------------------------------------------------------------------
def _meshgrid(size):                                               # piq.functional.get_meshgrid
    def axis(n):
        if n % 2:
            return torch.arange(-(n - 1) / 2, n / 2, dtype=torch.float64) / (n - 1)
        return torch.arange(-n / 2, n / 2, dtype=torch.float64) / n
    return torch.meshgrid(axis(size[0]), axis(size[1]), indexing="ij")


def _ifftshift(x):                                                 # piq.functional.ifftshift: roll by -(n // 2) on every axis
    return torch.roll(x, shifts=[-(n // 2) for n in x.shape], dims=list(range(x.dim())))


def fsim_filters(h, w, scales=4, orientations=4, min_length=6, mult=2, sigma_f=0.55, delta_theta=1.2):
    """[orientations * scales, h, w] log-Gabor filter bank of piq's _construct_filters (orientation-major)"""
    theta_sigma = math.pi / (orientations * delta_theta)
    gx, gy = _meshgrid((h, w))
    radius = torch.sqrt(gx ** 2 + gy ** 2)
    theta = torch.atan2(-gy, gx)
    radius, theta = _ifftshift(radius), _ifftshift(theta)
    radius[0, 0] = 1
    sintheta, costheta = torch.sin(theta), torch.cos(theta)
    lx, ly = _meshgrid((h, w))
    lp = _ifftshift(1.0 / (1.0 + (torch.sqrt(lx ** 2 + ly ** 2) / 0.45) ** (2 * 15)))        # _lowpassfilter(cutoff=.45, n=15)
    log_gabor = []
    for s_ in range(scales):
        omega_0 = 1.0 / (min_length * mult ** s_)
        g = torch.exp((-torch.log(radius / omega_0) ** 2) / (2 * math.log(sigma_f) ** 2)) * lp
        g[0, 0] = 0
        log_gabor.append(g)
    spread = []
    for o in range(orientations):
        angl = o * math.pi / orientations
        ds = sintheta * math.cos(angl) - costheta * math.sin(angl)
        dc = costheta * math.cos(angl) + sintheta * math.sin(angl)
        dtheta = torch.abs(torch.atan2(ds, dc))
        spread.append(torch.exp((-dtheta ** 2) / (2 * theta_sigma ** 2)))
    spread, log_gabor = torch.stack(spread), torch.stack(log_gabor)
    return spread.repeat_interleave(scales, dim=0) * log_gabor.repeat(orientations, 1, 1)


def _phase_congruency(x, scales=4, orientations=4, k=2.0, eps=torch.finfo(torch.float32).eps):
    """x: [N, 1, H, W] float64 -> [N, 1, H, W]; piq's _phase_congruency (EPS is float32's there: the reference runs in float32)"""
    n, _, h, w = x.shape
    filters = fsim_filters(h, w, scales, orientations).unsqueeze(0)
    imagefft = torch.fft.fft2(x)
    filters_ifft = torch.fft.ifft2(filters).real * math.sqrt(h * w)
    eo = torch.fft.ifft2(imagefft * filters).view(n, orientations, scales, h, w)
    even, odd = eo.real, eo.imag
    an = torch.sqrt(even ** 2 + odd ** 2)
    em_n = (filters.view(1, orientations, scales, h, w)[:, :, :1] ** 2).sum(dim=[-2, -1], keepdim=True)
    sum_e, sum_o = even.sum(dim=2, keepdim=True), odd.sum(dim=2, keepdim=True)
    x_energy = torch.sqrt(sum_e ** 2 + sum_o ** 2) + eps
    mean_e, mean_o = sum_e / x_energy, sum_o / x_energy
    energy = (even * mean_e + odd * mean_o - torch.abs(even * mean_o - odd * mean_e)).sum(dim=2, keepdim=True)
    abs_eo = an[:, :, :1].reshape(n, orientations, 1, 1, h * w)
    median_e2n = torch.median(abs_eo ** 2, dim=-1, keepdim=True).values          # lower median, like torch
    mean_e2n = -median_e2n / math.log(0.5)
    noise_power = mean_e2n / em_n
    fi = filters_ifft.view(1, orientations, scales, h, w)
    sum_an2 = (fi ** 2).sum(dim=-3, keepdim=True).sum(dim=[-1, -2], keepdim=True)
    est = torch.zeros(1, orientations, 1, h, w, dtype=x.dtype)
    for s_ in range(scales - 1):
        est = est + (fi[:, :, s_:s_ + 1] * fi[:, :, s_ + 1:]).sum(dim=-3, keepdim=True)
    sum_ai_aj = est.sum(dim=[-1, -2], keepdim=True)
    noise_energy2 = 2 * noise_power * sum_an2 + 4 * noise_power * sum_ai_aj
    tau = torch.sqrt(noise_energy2 / 2)
    t = (tau * math.sqrt(math.pi / 2) + k * torch.sqrt((2 - math.pi / 2) * tau ** 2)) / 1.7
    energy = torch.max(energy - t, torch.zeros_like(t))
    return ((energy.sum(dim=[1, 2]) + eps) / (an.sum(dim=[1, 2]) + eps)).unsqueeze(1)


def _similarity(a, b, c):                                          # piq.functional.similarity_map
    return (2.0 * a * b + c) / (a ** 2 + b ** 2 + c)


def fsim(x, y):
    """piq.fsim(x, y, reduction='none'-like: one value per sample), data_range 1, chromatic"""
    x, y = x.double() * 255, y.double() * 255
    f = metric_factor(x.shape[-2], x.shape[-1])
    x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    m = torch.tensor([[0.299, 0.587, 0.114], [0.5959, -0.2746, -0.3213], [0.2115, -0.5227, 0.3112]], dtype=torch.float64)   # rgb2yiq
    xq, yq = torch.einsum("kc,nchw->nkhw", m, x), torch.einsum("kc,nchw->nkhw", m, y)
    xl, yl = xq[:, :1], yq[:, :1]
    pc_x, pc_y = _phase_congruency(xl), _phase_congruency(yl)
    sch = torch.tensor([[-3.0, 0.0, 3.0], [-10.0, 0.0, 10.0], [-3.0, 0.0, 3.0]], dtype=torch.float64) / 16
    kern = torch.stack([sch, sch.t()]).unsqueeze(1)

    def grad(z):
        return torch.sqrt((F.conv2d(z, kern, padding=1) ** 2).sum(dim=1, keepdim=True))
    pc = _similarity(pc_x, pc_y, 0.85)
    gm = _similarity(grad(xl), grad(yl), 160.0)
    pc_max = torch.where(pc_x > pc_y, pc_x, pc_y)
    score = gm * pc * pc_max
    s_i, s_q = _similarity(xq[:, 1:2], yq[:, 1:2], 200.0), _similarity(xq[:, 2:], yq[:, 2:], 200.0)
    score = score * torch.abs(s_i * s_q) ** 0.03
    return score.sum(dim=[1, 2, 3]) / pc_max.sum(dim=[1, 2, 3])
