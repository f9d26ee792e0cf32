"""PAB parameter container -- same attribute tree as the reference's pasmnet/attention.py:9-16
(`head` ResB, `query`/`key`/`value` 1x1 convs).  The cost volumes of the reference's
PAB.forward (attention.py:39-46) are never materialised here: methods.dcmcs3di fuses
cost -> softmax -> warp in ct_pam_attend_f32 / ct_pam_valid_f32 (csrc/cnn.hip)."""
import torch

from pasmnet.backbone import ResB


class PAB(torch.nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.head = ResB(channels, channels)
        self.query = torch.nn.Conv2d(channels, channels, kernel_size=1)
        self.key = torch.nn.Conv2d(channels, channels, kernel_size=1)
        self.value = torch.nn.Conv2d(channels, channels, kernel_size=1)
