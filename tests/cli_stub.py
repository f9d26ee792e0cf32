"""Test stub for tests/test_cli_gloo.py: a model with the Runner's test_step interface whose arithmetic is plain torch on CPU
tensors (a per-frame channel-mean swap + the four metric names), so that utils.cli.main can be driven end to end -- argv,
YAML, sharding, per-frame metrics, the one gather, the printed means -- without a GPU.  Not product code."""
import torch


class StubRunner(torch.nn.Module):
    def __init__(self, func_spec="tests.cli_stub.swap_means", gain=1.0):
        super().__init__()
        self.func_spec, self.gain = func_spec, float(gain)

    def forward(self, batch):
        t, r = batch["target"], batch["reference"]
        return (t - t.mean(dim=(2, 3), keepdim=True)) * self.gain + r.mean(dim=(2, 3), keepdim=True)

    def test_step(self, batch, batch_idx=0, dataloader_idx=0):
        out = self(batch).clamp(0, 1)
        gt = batch["gt"]
        mse = ((out.double() - gt.double()) ** 2).flatten(1).mean(dim=1)
        # four deterministic per-frame numbers under the reference's metric names (only PSNR is the real formula)
        return {"Test PSNR": 10.0 * torch.log10(1.0 / mse), "Test SSIM": 1.0 - mse, "Test FSIM": out.double().flatten(1).mean(dim=1),
                "Test iCID": (out.double() - gt.double()).abs().flatten(1).max(dim=1).values}
