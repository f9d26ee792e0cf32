"""GMFlow / UniMatch optical-flow matcher -- MI355X drop-in for the FORWARD pass of the reference's
`unimatch.GMFlow` in the configuration DMSCT uses (methods/dmsct.py:85-94; unimatch/__init__.py:60-67:
task='flow', attn_type='swin', attn_splits_list=(2,8), corr_radius_list=(-1,4), prop_radius_list=(-1,1),
num_reg_refine=6, pred_bidir_flow=True).

Same parameter tree as the reference (`backbone.*`, `transformer.layers.N.*`, `feature_flow_attn.*`,
`refine_proj.*`, `refine.*`) so pretrained checkpoints `load_state_dict` strictly; no
network access: `pretrained` is None (random init) or a local checkpoint path.  All arithmetic runs in
HIP kernels (csrc/gmflow.hip); torch only allocates, concatenates and permutes.  No CPU fallback.
The stereo / depth branches of UniMatch are dead code for this repository and are not implemented.
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn

import ct_hip
from ct_hip import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH

default_cfg = {"num_scales": 2, "feature_channels": 128, "upsample_factor": 4, "num_head": 1, "ffn_dim_expansion": 4,
               "num_transformer_layers": 6, "reg_refine": True, "task": "flow"}


# ---- parameter containers (attribute names == reference) ---------------------------------------------------------
class ResidualBlock(nn.Module):                       # unimatch/backbone.py:9-31
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.stride = stride
        self.downsample = None
        if stride != 1 or in_planes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), nn.Identity())


class TridentWeight(nn.Module):                       # unimatch/trident_conv.py:50-61 (weight only, bias=False)
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, c, 3, 3))
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")


class CNNEncoder(nn.Module):                          # unimatch/backbone.py:42-102 with num_output_scales=2
    def __init__(self, output_dim=128):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, 1), ResidualBlock(64, 64, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, 2), ResidualBlock(96, 96, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, 1), ResidualBlock(128, 128, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.trident_conv = TridentWeight(output_dim)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class TransformerLayer(nn.Module):                    # unimatch/transformer.py:12-43
    def __init__(self, d=128, no_ffn=False, expansion=4):
        super().__init__()
        self.no_ffn = no_ffn
        self.q_proj, self.k_proj, self.v_proj = (nn.Linear(d, d, bias=False) for _ in range(3))
        self.merge = nn.Linear(d, d, bias=False)
        self.norm1 = nn.LayerNorm(d)
        if not no_ffn:
            self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d * expansion, bias=False), nn.GELU(),
                                     nn.Linear(2 * d * expansion, d, bias=False))
            self.norm2 = nn.LayerNorm(d)


class TransformerBlock(nn.Module):                    # unimatch/transformer.py:150-203
    def __init__(self, d=128, expansion=4):
        super().__init__()
        self.self_attn = TransformerLayer(d, no_ffn=True, expansion=expansion)
        self.cross_attn_ffn = TransformerLayer(d, expansion=expansion)


class FeatureTransformer(nn.Module):                  # unimatch/transformer.py:206-227
    def __init__(self, num_layers=6, d=128, expansion=4):
        super().__init__()
        self.layers = nn.ModuleList([TransformerBlock(d, expansion) for _ in range(num_layers)])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class SelfAttnPropagation(nn.Module):                 # unimatch/attention.py:169-186
    def __init__(self, c):
        super().__init__()
        self.q_proj, self.k_proj = nn.Linear(c, c), nn.Linear(c, c)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class FlowHead(nn.Module):                            # unimatch/reg_refine.py:9-22
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2 = nn.Conv2d(128, 256, 3, padding=1), nn.Conv2d(256, 2, 3, padding=1)


class SepConvGRU(nn.Module):                          # unimatch/reg_refine.py:25-39
    def __init__(self):
        super().__init__()
        for n in ("convz1", "convr1", "convq1"):
            setattr(self, n, nn.Conv2d(384, 128, (1, 5), padding=(0, 2)))
        for n in ("convz2", "convr2", "convq2"):
            setattr(self, n, nn.Conv2d(384, 128, (5, 1), padding=(2, 0)))


class BasicMotionEncoder(nn.Module):                  # unimatch/reg_refine.py:58-69
    def __init__(self):
        super().__init__()
        self.convc1, self.convc2 = nn.Conv2d(81, 256, 1), nn.Conv2d(256, 192, 3, padding=1)
        self.convf1, self.convf2 = nn.Conv2d(2, 128, 7, padding=3), nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(256, 126, 3, padding=1)


class BasicUpdateBlock(nn.Module):                    # unimatch/reg_refine.py:81-107
    def __init__(self):
        super().__init__()
        self.encoder, self.gru, self.flow_head = BasicMotionEncoder(), SepConvGRU(), FlowHead()
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 144, 1))


# ---- packed weights ----------------------------------------------------------------------------------------------
def _packed_conv(weight, bias):
    """ct_gconv2d_f32 operand layout of a conv's parameters, cached ON the weight Parameter and rebuilt when the
    parameters change (a global cache keyed by id() would serve recycled ids stale weights)."""
    ver = (weight._version, weight.data_ptr(), -1 if bias is None else bias._version,
           -1 if bias is None else bias.data_ptr(), str(weight.device))
    hit = getattr(weight, "_ct_packed", None)
    if hit is None or hit[0] != ver:
        hit = (ver, ct_hip.pack_gconv_weight(weight, bias))
        weight._ct_packed = hit
    return hit[1]


def _conv(m, x, act=ACT_NONE, stride=None, padding=None, weight=None, x2=None, out=None, s2d=None):
    """Conv2d module (or a bare weight) -> ct_gconv2d_f32; x2 = second input whose channels follow x's (no torch.cat); s2d =
    ct_hip.space_to_depth2(x) when several stride-2 convolutions read x"""
    w = m.weight if weight is None else weight
    b = getattr(m, "bias", None) if weight is None else None
    wp, bp = _packed_conv(w, b)
    st = (m.stride[0] if weight is None else 1) if stride is None else stride
    pd = (tuple(m.padding) if weight is None else 1) if padding is None else padding
    return ct_hip.gconv2d(x, wp, bp, w.shape[0], (w.shape[2], w.shape[3]), st, pd, act=act, x2=x2, out=out, s2d=s2d)


def _lin(m, x, act=ACT_NONE, x2=None):
    # the Parameter itself (not a detached alias): the pre-split bf16 image of the weight is cached on it
    return ct_hip.linear_tokens(x, m.weight, m.bias, act=act, x2=x2)


def _tokens(x):                                        # [B,C,H,W] -> [B,H*W,C]  (transformer.py:238-239)
    return ct_hip.nchw_to_tokens(x.contiguous())


def _nchw(t, h, w):                                    # [B,H*W,C] -> [B,C,H,W]
    return ct_hip.tokens_to_nchw(t.contiguous(), h, w)


# ---- host-side tables (float32 arithmetic identical to the reference's torch code) ---------------------------------
_tables = {}


def _position_table(h, w, feats, device):              # unimatch/position.py:26-46 for one window of h x w
    key = ("pos", h, w, feats, str(device))
    if key not in _tables:
        mask = torch.ones((1, h, w))
        y_embed, x_embed = mask.cumsum(1, dtype=torch.float32), mask.cumsum(2, dtype=torch.float32)
        eps, scale = 1e-6, 2 * math.pi
        y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
        x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
        dim_t = torch.arange(feats, dtype=torch.float32)
        dim_t = 10000 ** (2 * (dim_t // 2) / feats)
        pos_x, pos_y = x_embed[:, :, :, None] / dim_t, y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        _tables[key] = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).contiguous().to(device)   # [1,C,h,w]
    return _tables[key]


def _region_ids(h, w, splits, device):                 # unimatch/utils.py:87-105: region label of every token, window-major
    key = ("reg", h, w, splits, str(device))
    if key not in _tables:
        wh, ww = h // splits, w // splits
        sh, sw = wh // 2, ww // 2
        img = torch.zeros((h, w), dtype=torch.int32)
        cnt = 0
        for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
            for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
                img[hs, ws] = cnt
                cnt += 1
        reg = img.view(splits, wh, splits, ww).permute(0, 2, 1, 3).reshape(splits * splits, wh * ww).contiguous()
        _tables[key] = reg.to(device)
    return _tables[key]


def _window_rowmap(b, h, w, splits, shift, device):    # token rows of every (image, window), window-major
    key = ("rowmap", b, h, w, splits, bool(shift), str(device))
    if key not in _tables:
        wh, ww = h // splits, w // splits
        idx = torch.arange(b * h * w, dtype=torch.int32).view(b, h, w)
        if shift:
            idx = torch.roll(idx, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2))
        idx = idx.view(b, splits, wh, splits, ww).permute(0, 1, 3, 2, 4).reshape(b * splits * splits, wh * ww)
        _tables[key] = idx.contiguous().to(device)
    return _tables[key]


def _coords_tokens(b, h, w, device):                   # unimatch/geometry.py:8-25 as tokens [B, H*W, 2] (x, y)
    key = ("grid", b, h, w, str(device))
    if key not in _tables:
        y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        g = torch.stack([x, y], dim=-1).float().reshape(1, h * w, 2).repeat(b, 1, 1).contiguous()
        _tables[key] = g.to(device)
    return _tables[key]


class GMFlow(nn.Module):
    def __init__(self, pretrained=None, config=None):
        super().__init__()
        cfg = dict(default_cfg if config is None else config)
        if cfg != default_cfg:
            raise NotImplementedError("only the configuration of the pretrained GMFlow models is implemented")
        self.backbone = CNNEncoder(128)
        self.transformer = FeatureTransformer(6, 128, 4)
        self.feature_flow_attn = SelfAttnPropagation(128)
        self.refine_proj = nn.Conv2d(128, 256, 1)
        self.refine = BasicUpdateBlock()
        self.upsample_factor = 4
        if pretrained is not None:                     # a local path (the reference downloads from S3, __init__.py:10-13,55)
            ckpt = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt, strict=False)
        self.eval()

    # ---- unimatch/backbone.py:104-120 ----
    def _resblock(self, blk, x):
        # a down-sampling block's two stride-2 convolutions read one input: its space-to-depth image is made once
        s2d = ct_hip.space_to_depth2(x) if (blk.downsample is not None and blk.stride == 2 and ct_hip.s2d_ok(x)) else None
        y = ct_hip.instance_norm(_conv(blk.conv1, x, s2d=s2d), 1)
        y = _conv(blk.conv2, y)
        skip = x if blk.downsample is None else ct_hip.instance_norm(_conv(blk.downsample[0], x, s2d=s2d), 0)
        return ct_hip.instance_norm(y, 2, skip)         # relu(skip + relu(IN(conv2)))

    def _backbone(self, x):
        bb = self.backbone
        x = ct_hip.instance_norm(_conv(bb.conv1, x), 1)
        for layer in (bb.layer1, bb.layer2, bb.layer3):
            for blk in layer:
                x = self._resblock(blk, x)
        x = _conv(bb.conv2, x)
        w = bb.trident_conv.weight
        return [_conv(None, x, stride=1, padding=1, weight=w), _conv(None, x, stride=2, padding=1, weight=w)]

    # ---- unimatch/attention.py:48-107 on tokens ----
    @staticmethod
    def _window_attention(q, k, v, splits, shift, h, w, kv_swap=False):
        """split_feature / roll / merge_splits (attention.py:60-67,78-92,100-107) are one cached index table: the
        kernel gathers the tokens of a window and scatters its result through it, nothing is copied"""
        b = q.shape[0]
        rowmap = _window_rowmap(b, h, w, splits, shift, q.device)
        region = _region_ids(h, w, splits, q.device).repeat(b, 1).contiguous() if shift else None
        return ct_hip.attention_tokens(q, k, v, region, rowmap=rowmap, kv_shift=(q.shape[0] * q.shape[1]) // 2 if kv_swap else 0)

    def _tlayer(self, m, source, target, h, w, shift, splits, kv_swap=False):   # transformer.py:45-147
        """kv_swap: the layer's target is `target` with the two halves of the batch exchanged (the reference's concat1): the
        projections are per token, so they run on `target` as it is and the attention reads the keys / values of the other half"""
        # the projections reading the same tokens share one launch (feature slices of the resident-weight linear)
        if source is target:
            q, k, v = ct_hip.linear_tokens_multi(source, [m.q_proj.weight, m.k_proj.weight, m.v_proj.weight])
        else:
            q = _lin(m.q_proj, source)
            k, v = ct_hip.linear_tokens_multi(target, [m.k_proj.weight, m.v_proj.weight])
        att = self._window_attention(q, k, v, splits, shift, h, w, kv_swap)
        g1, b1 = m.norm1.weight.detach(), m.norm1.bias.detach()
        # merge projection + norm1 (+ the skip when the layer has no FFN) in one launch
        msg = ct_hip.linear_layernorm128(att, m.merge.weight, m.merge.bias, g1, b1, residual=source if m.no_ffn else None)
        if m.no_ffn:
            return msg
        hid = _lin(m.mlp[0], source, act=ACT_GELU, x2=msg)                     # mlp(cat([source, message])), no copy
        x = ct_hip.linear_tokens(hid, m.mlp[2].weight, m.mlp[2].bias, partials=True)
        slabs = x.shape[0] if x.dim() == hid.dim() + 1 else 1                  # K-sliced: partial slabs, summed by the LayerNorm
        return ct_hip.layernorm128(x, m.norm2.weight.detach(), m.norm2.bias.detach(), residual=source, partials=slabs)

    def _transformer(self, t0, t1, h, w, splits):                        # transformer.py:229-297
        # concat1 of the reference (transformer.py:281-287) = concat0 with the batch halves exchanged, re-built after every layer:
        # never materialised here -- the cross attention reads the other half's keys / values of the layer's input
        c0 = torch.cat((t0, t1), dim=0)
        for i, layer in enumerate(self.transformer.layers):
            shift = i % 2 == 1
            c_in = c0
            c0 = self._tlayer(layer.self_attn, c0, c0, h, w, shift, splits)
            c0 = self._tlayer(layer.cross_attn_ffn, c0, c_in, h, w, shift, splits, kv_swap=True)
        a, b_ = c0.chunk(2, dim=0)
        return a.contiguous(), b_.contiguous()

    def _add_position(self, f0, f1, splits):                             # utils.py:114-134
        b, c, h, w = f0.shape
        pos = _position_table(h // splits, w // splits, c // 2, f0.device).repeat(b, 1, splits, splits).contiguous()
        return ct_hip.eltwise(0, f0, pos), ct_hip.eltwise(0, f1, pos)

    def _gru_parts(self):
        """SepConvGRU weights split by input block, packed once (rebuilt when a parameter changes).  The convolutions read
        cat([h, inp, motion]) (reg_refine.py:43-55, 128 channels each); `inp` never changes over the refinement iterations and
        the hidden state entering the horizontal half is the loop-invariant `net` (unimatch.py:318-323 recomputes it every
        iteration), so   conv(cat) = [conv over the invariant blocks + bias: once per forward] + conv over the rest.
        name -> (packed invariant part, packed varying part, first invariant channel, invariant channels)"""
        gru = self.refine.gru
        names = [n + s for s in ("1", "2") for n in ("convz", "convr", "convq")]
        ver = tuple((getattr(gru, n).weight._version, getattr(gru, n).weight.data_ptr(), getattr(gru, n).bias._version,
                     getattr(gru, n).bias.data_ptr()) for n in names)
        hit = getattr(self, "_gru_packed", None)
        if hit is None or hit[0] != ver:
            parts = {}
            for n in names:
                w, b = getattr(gru, n).weight.detach(), getattr(gru, n).bias.detach()
                if n in ("convz1", "convr1"):              # h = net AND inp are invariant: only the motion block varies
                    fixed, rest = w[:, :256], w[:, 256:]
                else:                                      # h varies (r * h, or the horizontal half's output)
                    fixed, rest = w[:, 128:256], torch.cat((w[:, :128], w[:, 256:]), dim=1)
                parts[n] = (ct_hip.pack_gconv_weight(fixed.contiguous(), b),
                            ct_hip.pack_gconv_weight(rest.contiguous(), torch.zeros_like(b)))
            hit = (ver, parts)
            self._gru_packed = hit
        return hit[1]

    def _gru_invariants(self, net, inp):
        """the loop-invariant partial sums of the six GRU convolutions (bias included), computed once per forward"""
        gru, pre = self.refine.gru, {}
        for n, ((wp, bp), _) in self._gru_parts().items():
            conv = getattr(gru, n)
            ks, pd = tuple(conv.kernel_size), tuple(conv.padding)
            if n in ("convz1", "convr1"):
                pre[n] = ct_hip.gconv2d(net, wp, bp, 128, ks, 1, pd, x2=inp)
            else:
                pre[n] = ct_hip.gconv2d(inp, wp, bp, 128, ks, 1, pd)
        return pre

    def _refine_iter(self, net, x, corr, flow, want_mask, pre=None):     # reg_refine.py:58-122
        """The reference's torch.cat's (reg_refine.py:43,51,72,75,77) never materialise: convs read two tensors, and
        `x = [inp | motion features | flow]` is one buffer that `enc.conv` writes its 126 channels into.
        pre: _gru_invariants(net, inp) -- the GRU convolutions then run over the varying input blocks only."""
        enc, gru = self.refine.encoder, self.refine.gru
        cor = _conv(enc.convc2, _conv(enc.convc1, corr, ACT_RELU), ACT_RELU)
        flo = _conv(enc.convf2, _conv(enc.convf1, flow, ACT_RELU), ACT_RELU)
        _conv(enc.conv, cor, ACT_RELU, x2=flo, out=x[:, 128:254])       # x[:, :128] = inp is loop invariant (caller)
        x[:, 254:] = flow
        h = net
        if pre is not None:
            parts, mot = self._gru_parts(), x[:, 128:]

            fuse = ct_hip.conv_ws16()       # the fp16 form's epilogue also does the GRU's elementwise steps: r * h and the gate

            def part(name, a, act, x2=None, post=None):
                conv = getattr(gru, name)
                wp, bp = parts[name][1]
                return ct_hip.gconv2d(a, wp, bp, 128, tuple(conv.kernel_size), 1, tuple(conv.padding), act=act, x2=x2, addend=pre[name],
                                      post=post)
            for suf in ("1", "2"):
                hin = (mot, None) if suf == "1" else (h, mot)
                z = part("convz" + suf, hin[0], ACT_SIGMOID, x2=hin[1])
                if fuse:
                    rh = part("convr" + suf, hin[0], ACT_SIGMOID, x2=hin[1], post=(1, h, None))          # sigmoid(.) * h
                    h = part("convq" + suf, rh, ACT_TANH, x2=mot, post=(2, z, h))                        # (1 - z) h + z tanh(.)
                else:
                    r = part("convr" + suf, hin[0], ACT_SIGMOID, x2=hin[1])
                    q = part("convq" + suf, ct_hip.eltwise(1, r, h), ACT_TANH, x2=mot)
                    h = ct_hip.eltwise(2, z, h, q)
        else:
            for suf in ("1", "2"):
                z = _conv(getattr(gru, "convz" + suf), h, ACT_SIGMOID, x2=x)
                r = _conv(getattr(gru, "convr" + suf), h, ACT_SIGMOID, x2=x)
                q = _conv(getattr(gru, "convq" + suf), ct_hip.eltwise(1, r, h), ACT_TANH, x2=x)
                h = ct_hip.eltwise(2, z, h, q)
        delta = _conv(self.refine.flow_head.conv2, _conv(self.refine.flow_head.conv1, h, ACT_RELU))
        mask = _conv(self.refine.mask[2], _conv(self.refine.mask[0], h, ACT_RELU)) if want_mask else None
        return h, mask, delta

    @torch.no_grad()
    def _unimatch(self, img0, img1, num_reg_refine, dbg, bidir=True):    # unimatch.py:98-370 (fixed configuration)
        B = img0.shape[0]
        plane = img0.shape[2] * img0.shape[3]
        both = torch.cat((ct_hip.eltwise(3, img0, plane=plane), ct_hip.eltwise(3, img1, plane=plane)), dim=0)
        feats = self._backbone(both)[::-1]                                # low resolution first
        if dbg is not None:
            dbg["feat_s0"], dbg["feat_s1"] = feats[0], feats[1]
        flow = None
        for scale, (splits, corr_r, prop_r) in enumerate(((2, -1, -1), (8, 4, 1))):
            f0, f1 = feats[scale][:B], feats[scale][B:]
            if scale > 0 and bidir:
                f0, f1 = torch.cat((f0, f1), dim=0), torch.cat((f1, f0), dim=0)
            b, c, h, w = f0.shape
            f0_ori, f1_ori = f0, f1
            if scale > 0:
                flow = ct_hip.bilinear_resize(flow, (flow.shape[2] * 2, flow.shape[3] * 2), 2.0, 2.0)
                f1 = ct_hip.flow_warp(f1.contiguous(), flow)
            f0, f1 = self._add_position(f0.contiguous(), f1.contiguous(), splits)
            t0, t1 = self._transformer(_tokens(f0), _tokens(f1), h, w, splits)
            if dbg is not None:
                dbg["tf0_s%d" % scale] = _nchw(t0, h, w)
            if corr_r == -1:                                              # matching.py:10-39, both directions
                grid = _coords_tokens(b, h, w, f0.device)
                fwd = ct_hip.attention_tokens(t0, t1, grid)
                if bidir:
                    bwd = ct_hip.attention_tokens(t1, t0, grid)
                    corresp = torch.cat((fwd, bwd), dim=0)
                    grid2 = torch.cat((grid, grid), dim=0)
                else:                                                     # matching.py:27-39 without the transposed half
                    corresp, grid2 = fwd, grid
                pred = _nchw(ct_hip.eltwise(0, corresp, ct_hip.eltwise(4, grid2, s0=-1.0)), h, w)
            else:
                pred = ct_hip.local_corr_softmax(t0, t1, h, w, corr_r)
            flow = ct_hip.eltwise(0, flow, pred) if flow is not None else pred
            if dbg is not None:
                dbg["flow_match_s%d" % scale] = flow
            if scale == 0 and bidir:
                t0 = torch.cat((t0, t1), dim=0)
            prop = self.feature_flow_attn                                 # attention.py:188-256
            qtok = _lin(prop.q_proj, t0)
            if prop_r <= 0:
                ktok = _lin(prop.k_proj, qtok)                            # k_proj(q_proj(x)): the reference's documented quirk
                flow = _nchw(ct_hip.attention_tokens(qtok, ktok, _tokens(flow)), h, w)
            else:
                flow = ct_hip.local_attn_prop(qtok, _lin(prop.k_proj, t0), flow.contiguous(), prop_r)
            if dbg is not None:
                dbg["flow_prop_s%d" % scale] = flow
            if scale == 1:
                t0_ori, t1_ori = _tokens(f0_ori), _tokens(f1_ori)
                proj = ct_hip.eltwise(5, _conv(self.refine_proj, _nchw(t0, h, w)), plane=h * w, chans=256, split=128)
                net0 = proj[:, :128].contiguous()                                        # loop invariant (unimatch.py:318-323)
                xbuf = torch.empty((proj.shape[0], 256, h, w), dtype=torch.float32, device=proj.device)
                xbuf[:, :128] = proj[:, 128:]                                            # inp; [128:254] motion features, [254:] flow
                # split mode: the invariant input blocks of the GRU convolutions are convolved once (8 of the 18 block
                # convolutions of an iteration); exact mode keeps the reference's form
                pre = self._gru_invariants(net0, xbuf[:, :128]) if ct_hip.conv_mode() == "split" and w % 4 == 0 else None
                for it in range(num_reg_refine):
                    corr = ct_hip.local_corr_flow(t0_ori, t1_ori, flow, 4)
                    last = it == num_reg_refine - 1
                    _, up_mask, dflow = self._refine_iter(net0, xbuf, corr, flow, want_mask=last, pre=pre)
                    flow = ct_hip.eltwise(0, flow, dflow)
                    if dbg is not None:
                        dbg["flow_refine_%d" % it] = flow
                return ct_hip.convex_upsample(flow, up_mask, self.upsample_factor)

    @torch.no_grad()
    def forward(self, img0, img1, padding_factor=32, inference_size=None, attn_type="swin", attn_splits_list=(2, 8),
                corr_radius_list=(-1, 4), prop_radius_list=(-1, 1), num_reg_refine=6, pred_bidir_flow=False,
                pred_bwd_flow=False, pred_flow_viz=False, fwd_bwd_consistency_check=False, dbg=None, **kwargs):
        """unimatch/__init__.py:60-167 with the default split / radius lists: DMSCT's call (pred_bidir_flow=True, occlusion masks)
        and the one-direction forms (pred_bidir_flow=False; pred_bwd_flow swaps the frames, :117-118).  No flow visualisation."""
        if (pred_flow_viz or attn_type != "swin" or tuple(attn_splits_list) != (2, 8)
                or tuple(corr_radius_list) != (-1, 4) or tuple(prop_radius_list) != (-1, 1)):
            raise NotImplementedError("GMFlow on HIP implements the default split / radius lists, without pred_flow_viz")
        if fwd_bwd_consistency_check and not pred_bidir_flow:
            raise AssertionError("fwd_bwd_consistency_check needs pred_bidir_flow (unimatch/__init__.py:85-86)")
        if not img0.is_cuda:
            raise ct_hip.CtHipError("GMFlow runs on the GPU only (no CPU fallback)")
        img0, img1 = img0.float().contiguous(), img1.float().contiguous()
        transpose = img0.size(-2) > img0.size(-1)
        if transpose:
            img0, img1 = img0.transpose(-2, -1).contiguous(), img1.transpose(-2, -1).contiguous()
        nearest = [int(np.ceil(img0.size(-2) / padding_factor)) * padding_factor,
                   int(np.ceil(img0.size(-1) / padding_factor)) * padding_factor]
        size = nearest if inference_size is None else list(inference_size)
        ori = (img0.shape[-2], img0.shape[-1])
        resize = size[0] != ori[0] or size[1] != ori[1]
        if resize:
            img0, img1 = ct_hip.bilinear_resize(img0, size), ct_hip.bilinear_resize(img1, size)
        if pred_bwd_flow:
            img0, img1 = img1, img0
        flow = self._unimatch(img0, img1, num_reg_refine, dbg, bidir=bool(pred_bidir_flow))
        if resize:
            flow = ct_hip.bilinear_resize(flow, ori, ori[1] / size[1], ori[0] / size[0])
        if transpose:
            flow = flow.transpose(-2, -1).contiguous()
        if not pred_bidir_flow:
            return {"flow": flow}
        fwd, bwd = flow[::2].contiguous(), flow[1::2].contiguous()
        res = {"flow": fwd, "flow_bwd": bwd}
        if fwd_bwd_consistency_check:
            fo, bo = ct_hip.fb_check(fwd, bwd)
            res.update({"fwd_occ": fo.unsqueeze(1), "bwd_occ": bo.unsqueeze(1)})
        return res
