"""`Runner` -- the plugin loader of the reference (methods/__init__.py:10-40) without Lightning.

`Runner(func_spec)` resolves a dotted path "pkg.mod.func" with importlib exactly like the reference
(methods/__init__.py:14-16); `forward(batch)` applies it to every sample of the batch
(methods/__init__.py:18-27) and `test_step` clamps and scores the result (methods/__init__.py:29-40):
PSNR, SSIM, FSIM and iCID per frame on the GPU (ct_frame_*_f32; piq / kornia / torchvision arithmetic restated, see
oracle/metrics.py).

If the resolved module also offers a device-resident variant `<func>_cuda`, the batch never leaves
the GPU; otherwise the numpy callable is used through the same host round trip as the reference.
"""
import importlib

import torch


def psnr(x, y, data_range=1.0):
    """Peak signal-to-noise ratio per sample, [B] (piq.psnr semantics: mean over C,H,W, then 10 log10(1/mse)); the
    reduction runs in ct_frame_psnr_f32 (deterministic float64 sums) when the tensors live on the GPU."""
    if x.is_cuda and data_range == 1.0:
        import ct_hip
        return ct_hip.frame_psnr(x.float().contiguous(), y.float().contiguous())[:, 1]
    mse = ((x.double() - y.double()) ** 2).flatten(1).mean(dim=1)
    return 10.0 * torch.log10(data_range ** 2 / mse.clamp_min(1e-300))


def ssim(x, y):
    """piq.ssim(x, y) with piq's defaults, per sample [B] (GPU only: ct_frame_ssim_f32)."""
    import ct_hip
    return ct_hip.frame_ssim(x.float().contiguous(), y.float().contiguous())


def icid(x, y):
    """utils.icid.icid(x, y) (perceptual intent), per sample [B] (GPU only: ct_frame_icid_f32).  The reference returns
    the mean over the batch; test loaders use batch size 1 (utils/data.py:168-179), where the two coincide."""
    import ct_hip
    return ct_hip.frame_icid(x.float().contiguous(), y.float().contiguous())


def fsim(x, y):
    """piq.fsim(x, y) with piq's defaults (chromatic), per sample [B] (GPU only: ct_frame_fsim_f32)."""
    import ct_hip
    return ct_hip.frame_fsim(x.float().contiguous(), y.float().contiguous())


METRICS = ("Test PSNR", "Test SSIM", "Test FSIM", "Test iCID")        # what test_step logs, in its order (methods/__init__.py:37-40)


class Runner(torch.nn.Module):
    def __init__(self, func_spec, metrics=None):
        """func_spec: the reference's only argument (methods/__init__.py:12).  metrics (not in the reference): which of psnr, ssim,
        fsim, icid test_step computes -- default all four, like the reference; `psnr` alone lets the Reinhard transfer take its
        fused uint8 entry (test_group)."""
        super().__init__()
        specs = func_spec.split(".")
        module, func = ".".join(specs[:-1]), specs[-1]
        mod = importlib.import_module(module)
        self.func = getattr(mod, func)
        self.func_cuda = getattr(mod, func + "_cuda", None)
        self.func_spec = func_spec
        names = ("psnr", "ssim", "fsim", "icid") if metrics is None else tuple(m.strip() for m in (metrics.split(",") if isinstance(metrics, str) else metrics))
        if not names or any(m not in ("psnr", "ssim", "fsim", "icid") for m in names):
            raise ValueError("metrics: a subset of psnr, ssim, fsim, icid (got %r)" % (metrics,))
        self.metrics = names
        self._out = None

    def takes_groups(self):
        """True when whole groups of uint8 frames can go through ONE fused call: the Reinhard transfer with PSNR as the only
        metric (ct_reinhard_psnr_u8 reads the bytes; k / 255 and its gamma expansion come from tables inside the kernel)."""
        return self.func_spec == "methods.linear.color_transfer_between_images" and self.metrics == ("psnr",)

    def test_group(self, group_u8, psnr_out):
        """group_u8: device uint8 [3 roles: target, reference, gt][k][H][W][3]; psnr_out: float64 [k, 2] rows of the caller's table,
        filled with (mse, PSNR) per frame -- Runner.test_step's clamp + piq.psnr (methods/__init__.py:30-32) for k frames in one
        asynchronous call, nothing else on the stream."""
        import ct_hip
        k = group_u8.shape[1]
        if self._out is None or self._out.shape[1:] != group_u8.shape[2:] or self._out.shape[0] < k or self._out.device != group_u8.device:
            self._out = torch.empty((k,) + tuple(group_u8.shape[2:]), dtype=torch.float32, device=group_u8.device)
        ct_hip.reinhard_persist(group_u8[0], group_u8[1], gt=group_u8[2], out=self._out[:k], psnr_out=psnr_out)
        return self._out[:k]

    def forward(self, batch):
        target, reference = batch["target"], batch["reference"]
        if self.func_cuda is not None and target.is_cuda:
            # [B,3,H,W] -> HWC on the device; one call per sample like the reference's loop (so that e.g. IDT draws
            # fresh rotations per sample exactly as methods/__init__.py:20-25 + iterative.py:32 do)
            t = target.permute(0, 2, 3, 1).contiguous()
            r = reference.permute(0, 2, 3, 1).contiguous()
            out = torch.stack([self.func_cuda(a, b) for a, b in zip(t, r)])
            return out.float().permute(0, 3, 1, 2)
        outputs = []
        for t, r in zip(target, reference):
            t = t.permute(1, 2, 0).detach().cpu().numpy()
            r = r.permute(1, 2, 0).detach().cpu().numpy()
            outputs.append(torch.from_numpy(self.func(t, r)).float().permute(2, 0, 1))
        return torch.stack(outputs).to(target.device)

    def test_step(self, batch, batch_idx=0, dataloader_idx=0):
        result = self(batch).clamp(0, 1)
        gt = batch["gt"].to(result.device)
        out = {}
        fns = {"psnr": ("Test PSNR", psnr), "ssim": ("Test SSIM", ssim), "fsim": ("Test FSIM", fsim), "icid": ("Test iCID", icid)}
        for m in self.metrics:
            if m == "psnr" or result.is_cuda:        # SSIM / FSIM / iCID exist on the GPU only
                out[fns[m][0]] = fns[m][1](result, gt)
        return out
