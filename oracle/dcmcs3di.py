"""DCMCS3DI forward, CPU oracle: a functional float64 restatement of the reference's
methods/dcmcs3di.py:53-66, pasmnet/backbone.py:14-15, pasmnet/attention.py:33-48,
pasmnet/utils.py:28-40,123-125 that works from a plain state_dict (no module classes).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import torch
import torch.nn.functional as F


def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"].double(), sd[name + ".bias"].double(), padding=pad)


def _resb(sd, name, x):
    t = F.leaky_relu(_conv(sd, name + ".body.0", x, 1), 0.01)
    return x + _conv(sd, name + ".body.2", t, 1)


def forward(sd, left, right, extraction_layers=18, transfer_layers=6, valid_override=None):
    """Returns a dict with the same intermediates the goldens hold. sd: name -> tensor.
    valid_override: a boolean [B,1,H,W] mask used in place of `colsum > 0.1` (the threshold is discontinuous: a test that
    wants to compare the arithmetic behind it feeds both sides the same mask); adds `pre_clamp_override` to the result."""
    left, right = left.double(), right.double()

    def extraction(x):
        x = _conv(sd, "extraction.0", x, 1)
        for i in range(1, extraction_layers + 1):
            x = _resb(sd, "extraction.%d" % i, x)
        return x

    fea_left, fea_right = extraction(left), extraction(right)
    c = fea_left.shape[1]
    hl, hr = _resb(sd, "matcher.head", fea_left), _resb(sd, "matcher.head", fea_right)
    # cost_right2left = Q(left) K(right) / c ; cost_left2right = Q(right) K(left) / c
    Q = _conv(sd, "matcher.query", hl, 0).permute(0, 2, 3, 1)
    K = _conv(sd, "matcher.key", hr, 0).permute(0, 2, 1, 3)
    cost_r2l = torch.matmul(Q, K) / c
    Q = _conv(sd, "matcher.query", hr, 0).permute(0, 2, 3, 1)
    K = _conv(sd, "matcher.key", hl, 0).permute(0, 2, 1, 3)
    cost_l2r = torch.matmul(Q, K) / c
    att_r2l = F.softmax(cost_r2l, dim=-1)
    att_l2r = F.softmax(cost_l2r, dim=-1)
    colsum = att_l2r.sum(dim=-2)
    valid_left = (colsum > 0.1).unsqueeze(1)

    def warp(img, att):
        return torch.matmul(att, img.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

    fea_warped = warp(_conv(sd, "matcher.value", fea_right, 0), att_r2l)

    def transfer(valid):
        x = torch.cat([fea_left, fea_warped, valid.double()], dim=1)
        x = _conv(sd, "transfer.0", x, 0)
        for i in range(1, transfer_layers + 1):
            x = _resb(sd, "transfer.%d" % i, x)
        x = _conv(sd, "transfer.%d" % (transfer_layers + 1), x, 1)
        return _conv(sd, "transfer.%d" % (transfer_layers + 2), x, 1)

    pre = transfer(valid_left)
    extra = {} if valid_override is None else {"pre_clamp_override": transfer(valid_override)}
    return dict(extra, fea_left=fea_left, fea_right=fea_right, cost_r2l=cost_r2l, cost_l2r=cost_l2r, att_r2l=att_r2l,
                att_l2r=att_l2r, colsum=colsum, valid_left=valid_left, fea_warped=fea_warped, pre_clamp=pre,
                corrected=pre.clamp(0, 1), warped_rgb=warp(right, att_r2l))
