"""ResB parameter container -- same attribute tree as the reference's pasmnet/backbone.py:4-15
(`body.0` conv3x3, `body.1` LeakyReLU, `body.2` conv3x3), so reference checkpoints load strictly.
The arithmetic runs in ct_conv2d_f32 (csrc/cnn.hip); there is no eager/CPU forward."""
import torch


class ResB(torch.nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.body = torch.nn.Sequential(
            torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            torch.nn.LeakyReLU(inplace=True),
            torch.nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
        )

    def forward(self, x):
        from methods.dcmcs3di import resb_forward
        return resb_forward(self, x)
