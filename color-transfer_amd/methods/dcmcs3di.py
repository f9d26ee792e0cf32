"""Deep Color Mismatch Correction in Stereoscopic 3D Images (Croci et al. 2021) -- MI355X drop-in
for the FORWARD pass of the reference's methods/dcmcs3di.py:29-66.

Same constructor arguments, parameter names/shapes (reference checkpoints `load_state_dict`
strictly) and `forward(left, right, inference)` return structure.  The ResB convolutions run in
ct_conv3x3_ws16_f32 (weights stationary in registers, two fp16 pieces per float32 operand), the
other convolutions in ct_conv2d_split_f32 (three bf16 pieces) -- or all of them in ct_conv2d_f32
(exact-f32 MFMA) in `exact` mode; the parallax attention in the streaming kernels behind
ct_hip.pam_streaming (ct_pam_attend_f32 / ct_pam_valid_f32 when the [B,H,W,W] maps are wanted); torch
only allocates tensors.  Training (`step`, losses, logging: dcmcs3di.py:68-147) is out of scope.
No CPU fallback.
"""
import torch

import ct_hip
from pasmnet.attention import PAB
from pasmnet.backbone import ResB, conv_forward, conv_forward_rows, resb_forward


def sequential_forward(seq, x):
    for m in seq:
        if isinstance(m, ResB):
            x = resb_forward(m, x)
        elif isinstance(m, torch.nn.Conv2d):
            x = conv_forward(m, x)
        else:
            raise TypeError("unexpected module %r" % (m,))
    return x


class DCMCS3DI(torch.nn.Module):
    def __init__(self, extraction_layers=18, transfer_layers=6, channels=64):
        super().__init__()
        self.hparams = type("HParams", (), dict(extraction_layers=extraction_layers, transfer_layers=transfer_layers,
                                                channels=channels))()
        # the reference takes any width (dcmcs3di.py:30-51).  64 (configs/dcmcs3di.yaml) runs on the kernels built for it (Winograd ResB
        # convolutions, streaming attention at any image width); other widths take the generic paths: tile convolutions, and the
        # LDS-tile attention (images up to ~1000 columns).  The three-source 1x1 convolution of transfer[0] wants channel counts that are
        # multiples of 16; the attention kernels contract over 64 query / key channels (narrower ones are zero-padded, see forward_parts).
        if channels % 16 or not 16 <= channels <= 64:
            raise ValueError("channels must be 16, 32, 48 or 64 (the reference's configs/dcmcs3di.yaml uses 64)")
        # construction order == reference (dcmcs3di.py:41-51): torch.manual_seed(s) gives the same init
        self.extraction = torch.nn.Sequential(torch.nn.Conv2d(3, channels, kernel_size=3, padding=1))
        for _ in range(extraction_layers):
            self.extraction.append(ResB(channels, channels))
        self.matcher = PAB(channels)
        self.transfer = torch.nn.Sequential(torch.nn.Conv2d(2 * channels + 1, channels, kernel_size=1))
        for _ in range(transfer_layers):
            self.transfer.append(ResB(channels, channels))
        self.transfer.append(torch.nn.Conv2d(channels, channels // 2, kernel_size=3, padding=1))
        self.transfer.append(torch.nn.Conv2d(channels // 2, 3, kernel_size=3, padding=1))

    @torch.no_grad()
    def forward_parts(self, left, right, want_att=False, want_valid_right=False):
        """The forward pass with its intermediates (used by forward() and by the parity tests)."""
        if not left.is_cuda:
            raise ct_hip.CtHipError("DCMCS3DI runs on the GPU only (no CPU fallback)")
        left = left.contiguous().float()
        right = right.contiguous().float()
        B = left.shape[0]
        both = torch.cat([left, right], dim=0)
        fea = sequential_forward(self.extraction, both)                    # dcmcs3di.py:54-55
        head = resb_forward(self.matcher.head, fea)                        # attention.py:35-36
        fea_left, fea_right = fea[:B], fea[B:]
        H, W = left.shape[2], left.shape[3]
        tile_path = want_att or self.hparams.channels != 64      # the streaming kernels are built for 64 channels
        if tile_path or want_valid_right:
            q = conv_forward(self.matcher.query, head)                     # attention.py:39,44
            k = conv_forward(self.matcher.key, head)                       # attention.py:40,45
            c = q.shape[1]
            if c != 64:
                # the attention kernels contract over 64 channels and scale the scores by 1 / 64; the reference scales by 1 / c
                # (attention.py:41): zero channels add nothing to a score, and q * (64 / c) puts the scale right
                q64, k64 = q.new_zeros((q.shape[0], 64) + tuple(q.shape[2:])), k.new_zeros((k.shape[0], 64) + tuple(k.shape[2:]))
                q64[:, :c] = q * (64.0 / c)
                k64[:, :c] = k
                q, k = q64, k64
        if tile_path:
            # the [B,H,W,W] maps are wanted (or the width is not 64): LDS-tile kernels that can write them out
            v = conv_forward(self.matcher.value, fea_right)                # dcmcs3di.py:58
            # right-to-left: Q(left) . K(right)
            fea_warped, warped_rgb, att_r2l = ct_hip.pam_attend(q[:B].contiguous(), k[B:].contiguous(), v, right, want_att=want_att)
            # left-to-right softmax, column sums -> valid mask of the LEFT view (utils.py:31,34-35)
            valid_left, colsum_left, att_l2r = ct_hip.pam_valid(q[B:].contiguous(), k[:B].contiguous(), want_att=want_att)
        else:
            # streaming (online-softmax) kernels: no score tile in LDS, any width, K/V rows fetched with 16-byte loads.  They read
            # token rows [B*H, W, C]; the three 1x1 convolutions store that layout from their epilogues (no NCHW q/k/v, no
            # transposes)
            att_r2l = att_l2r = None
            qt = conv_forward_rows(self.matcher.query, head)               # [2B*H, W, 64]: Q(left) rows then Q(right) rows
            kt = conv_forward_rows(self.matcher.key, head)
            vt = conv_forward_rows(self.matcher.value, fea_right, channels=96)
            n = B * H
            fea_warped, warped_rgb, valid_left, colsum_left = ct_hip.pam_streaming_rows(qt[:n], kt[n:], vt, right, qt[n:], kt[:n])
        # dcmcs3di.py:59,47: transfer[0] (1x1, 129 -> 64) reads cat([fea_left, fea_warped, valid_left]) straight from its
        # three tensors (a three-source K loop in ct_conv2d_split_f32; the 129-channel tensor is never built)
        x = conv_forward(self.transfer[0], fea_left, x2=fea_warped, x3=valid_left)
        n_t = len(self.transfer)
        for i in range(1, n_t - 1):
            m = self.transfer[i]
            x = resb_forward(m, x) if isinstance(m, ResB) else conv_forward(m, x)
        pre_clamp = conv_forward(self.transfer[n_t - 1], x)
        parts = dict(fea_left=fea_left, fea_right=fea_right, fea_warped=fea_warped, warped_rgb=warped_rgb,
                     att_r2l=att_r2l, att_l2r=att_l2r, valid_left=valid_left, colsum_left=colsum_left,
                     pre_clamp=pre_clamp, corrected=pre_clamp.clamp(min=0, max=1))
        if want_valid_right:
            parts["valid_right"], parts["colsum_right"], _ = ct_hip.pam_valid(q[:B], k[B:])
        return parts

    def forward(self, left, right, inference=False, return_attention=None):
        """Same return structure as the reference (dcmcs3di.py:61-66):
        (corrected, ((att_r2l, att_l2r), (cycle_l, cycle_r), (valid_left, valid_right), warp(right, att_r2l))).
        The [B,H,W,W] attention maps (15.9 GB each at 1080p) are materialised only when
        `return_attention` is true (default: only in training mode, inference=False)."""
        want_att = (not inference) if return_attention is None else bool(return_attention)
        p = self.forward_parts(left, right, want_att=want_att, want_valid_right=not inference)
        valid_left = p["valid_left"] > 0.5
        if inference:
            att_cycle = (None, None)
            valid = (valid_left, None)
        else:
            att_cycle = (torch.matmul(p["att_r2l"], p["att_l2r"]), torch.matmul(p["att_l2r"], p["att_r2l"])) \
                if want_att else (None, None)
            valid = (valid_left, p["valid_right"] > 0.5)
        return p["corrected"], ((p["att_r2l"], p["att_l2r"]), att_cycle, valid, p["warped_rgb"])
