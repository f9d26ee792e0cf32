"""GMFlow / UniMatch matcher as called by DMSCT (reference methods/dmsct.py:85-94), CPU oracle.

A functional torch restatement of the ONE configuration the reference uses
(unimatch/__init__.py:60-67: task='flow', attn_type='swin', attn_splits_list=(2,8),
corr_radius_list=(-1,4), prop_radius_list=(-1,1), num_reg_refine=6, pred_bidir_flow=True,
fwd_bwd_consistency_check=True), written from a plain state_dict -- no module classes of the
reference are imported.  Every function cites the reference lines it follows.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  `dtype` selects float32 (default, what the
reference computes in) or float64.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---- unimatch/geometry.py -------------------------------------------------------------------------
def coords_grid(b, h, w, dtype):                                   # geometry.py:8-25
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], dim=0).to(dtype)[None].repeat(b, 1, 1, 1)


def bilinear_sample(img, coords):                                  # geometry.py:43-65 (zeros padding, align_corners)
    b, _, h, w = coords.shape
    xg = 2 * coords[:, 0] / (w - 1) - 1
    yg = 2 * coords[:, 1] / (h - 1) - 1
    return F.grid_sample(img, torch.stack([xg, yg], dim=-1), mode="bilinear", padding_mode="zeros", align_corners=True)


def flow_warp(feature, flow):                                      # geometry.py:68-75
    b, c, h, w = feature.shape
    return bilinear_sample(feature, coords_grid(b, h, w, flow.dtype) + flow)


def forward_backward_consistency_check(fwd, bwd, alpha=0.01, beta=0.5):   # geometry.py:78-99
    mag = torch.norm(fwd, dim=1) + torch.norm(bwd, dim=1)
    diff_fwd = torch.norm(fwd + flow_warp(bwd, fwd), dim=1)
    diff_bwd = torch.norm(bwd + flow_warp(fwd, bwd), dim=1)
    thr = alpha * mag + beta
    return (diff_fwd > thr).to(fwd.dtype), (diff_bwd > thr).to(fwd.dtype)


# ---- unimatch/utils.py, position.py ------------------------------------------------------------------
def split_feature(x, k):                                           # utils.py:37-63 (channel first)
    b, c, h, w = x.shape
    return x.view(b, c, k, h // k, k, w // k).permute(0, 2, 4, 1, 3, 5).reshape(b * k * k, c, h // k, w // k)


def merge_splits(x, k):                                            # utils.py:66-84 (channel first)
    b, c, h, w = x.shape
    nb = b // k // k
    return x.view(nb, k, k, c, h, w).permute(0, 3, 1, 4, 2, 5).contiguous().view(nb, c, k * h, k * w)


def position_embedding_sine(b, h, w, num_pos_feats, dtype):        # position.py:26-46 (float32 arithmetic like the reference)
    mask = torch.ones((b, h, w))
    y_embed = mask.cumsum(1, dtype=torch.float32)
    x_embed = mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).to(dtype)


def feature_add_position(f0, f1, splits, channels):                # utils.py:114-134
    f0s, f1s = split_feature(f0, splits), split_feature(f1, splits)
    pos = position_embedding_sine(f0s.shape[0], f0s.shape[2], f0s.shape[3], channels // 2, f0.dtype)
    return merge_splits(f0s + pos, splits), merge_splits(f1s + pos, splits)


def shift_window_mask(h, w, wh, ww):                               # utils.py:87-111
    sh, sw = wh // 2, ww // 2
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    k = w // ww
    mw = img.view(1, k, h // k, k, w // k, 1).permute(0, 1, 3, 2, 4, 5).reshape(k * k, wh * ww)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def upsample_flow_with_mask(flow, mask, factor):                   # utils.py:137-155
    b, c, h, w = flow.shape
    mask = torch.softmax(mask.view(b, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(b, c, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(b, c, factor * h, factor * w)


# ---- unimatch/backbone.py, trident_conv.py ---------------------------------------------------------------
def _inorm(x):                                                     # nn.InstanceNorm2d defaults (backbone.py:10,44,54)
    return F.instance_norm(x, eps=1e-5)


def residual_block(sd, pre, x, stride):                            # backbone.py:9-39
    y = F.relu(_inorm(F.conv2d(x, sd[pre + ".conv1.weight"], None, stride=stride, padding=1)))
    y = F.relu(_inorm(F.conv2d(y, sd[pre + ".conv2.weight"], None, padding=1)))
    if pre + ".downsample.0.weight" in sd:
        x = _inorm(F.conv2d(x, sd[pre + ".downsample.0.weight"], sd[pre + ".downsample.0.bias"], stride=stride))
    return F.relu(x + y)


def cnn_encoder(sd, x):                                            # backbone.py:104-120 with num_output_scales = 2
    x = F.relu(_inorm(F.conv2d(x, sd["backbone.conv1.weight"], None, stride=2, padding=3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
        x = residual_block(sd, "backbone.%s.0" % name, x, stride)
        x = residual_block(sd, "backbone.%s.1" % name, x, 1)
    x = F.conv2d(x, sd["backbone.conv2.weight"], sd["backbone.conv2.bias"])
    w = sd["backbone.trident_conv.weight"]                         # trident_conv.py:64-72: shared weight, strides (1, 2)
    return [F.conv2d(x, w, None, stride=1, padding=1), F.conv2d(x, w, None, stride=2, padding=1)]


# ---- unimatch/attention.py, transformer.py ----------------------------------------------------------------
def split_window_attention(q, k, v, splits, with_shift, h, w, mask):   # attention.py:48-107
    b, _, c = q.shape
    wh, ww = h // splits, w // splits
    q, k, v = (t.view(b, h, w, c) for t in (q, k, v))
    if with_shift:
        q, k, v = (torch.roll(t, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2)) for t in (q, k, v))

    def sp(t):
        return t.view(b, splits, wh, splits, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(b * splits * splits, wh * ww, c)
    q, k, v = sp(q), sp(k), sp(v)
    scores = torch.matmul(q, k.permute(0, 2, 1)) / c ** 0.5
    if with_shift:
        scores = scores + mask.repeat(b, 1, 1)
    out = torch.matmul(torch.softmax(scores, dim=-1), v)
    out = out.view(b, splits, splits, wh, ww, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, c)
    if with_shift:
        out = torch.roll(out, shifts=(wh // 2, ww // 2), dims=(1, 2))
    return out.view(b, -1, c)


def transformer_layer(sd, pre, source, target, h, w, mask, with_shift, splits, ffn):   # transformer.py:45-147
    q = F.linear(source, sd[pre + ".q_proj.weight"])
    k = F.linear(target, sd[pre + ".k_proj.weight"])
    v = F.linear(target, sd[pre + ".v_proj.weight"])
    msg = split_window_attention(q, k, v, splits, with_shift, h, w, mask)
    msg = F.layer_norm(F.linear(msg, sd[pre + ".merge.weight"]), (q.shape[-1],), sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"])
    if ffn:
        x = torch.cat([source, msg], dim=-1)
        x = F.linear(F.gelu(F.linear(x, sd[pre + ".mlp.0.weight"])), sd[pre + ".mlp.2.weight"])
        msg = F.layer_norm(x, (q.shape[-1],), sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"])
    return source + msg


def feature_transformer(sd, f0, f1, splits, num_layers=6):         # transformer.py:229-297
    b, c, h, w = f0.shape
    t0, t1 = f0.flatten(-2).permute(0, 2, 1), f1.flatten(-2).permute(0, 2, 1)
    mask = shift_window_mask(h, w, h // splits, w // splits).to(f0.dtype)
    c0, c1 = torch.cat((t0, t1), dim=0), torch.cat((t1, t0), dim=0)
    for i in range(num_layers):
        pre = "transformer.layers.%d" % i
        shift = i % 2 == 1
        c0 = transformer_layer(sd, pre + ".self_attn", c0, c0, h, w, mask, shift, splits, ffn=False)
        c0 = transformer_layer(sd, pre + ".cross_attn_ffn", c0, c1, h, w, mask, shift, splits, ffn=True)
        c1 = torch.cat(c0.chunk(2, dim=0)[::-1], dim=0)
    t0, t1 = c0.chunk(2, dim=0)
    return (t0.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous(), t1.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous())


def self_attn_propagation(sd, f0, flow, local_radius):             # attention.py:169-256
    b, c, h, w = f0.shape
    wq, bq = sd["feature_flow_attn.q_proj.weight"], sd["feature_flow_attn.q_proj.bias"]
    wk, bk = sd["feature_flow_attn.k_proj.weight"], sd["feature_flow_attn.k_proj.bias"]
    tokens = f0.view(b, c, h * w).permute(0, 2, 1)
    if local_radius <= 0:                                          # global: k = k_proj(q_proj(x)) (attention.py:201-208)
        q = F.linear(tokens, wq, bq)
        k = F.linear(q, wk, bk)
        v = flow.view(b, flow.shape[1], h * w).permute(0, 2, 1)
        prob = torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / c ** 0.5, dim=-1)
        return torch.matmul(prob, v).view(b, h, w, v.shape[-1]).permute(0, 3, 1, 2)
    ks = 2 * local_radius + 1                                      # local: k = k_proj(x) (attention.py:220-256)
    q = F.linear(tokens, wq, bq).reshape(b * h * w, 1, c)
    kp = F.linear(tokens, wk, bk).permute(0, 2, 1).reshape(b, c, h, w)
    kw = F.unfold(kp, kernel_size=ks, padding=local_radius).view(b, c, ks * ks, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, c, ks * ks)
    vw = F.unfold(flow, kernel_size=ks, padding=local_radius).view(b, flow.shape[1], ks * ks, h, w).permute(0, 3, 4, 2, 1).reshape(b * h * w, ks * ks, flow.shape[1])
    prob = torch.softmax(torch.matmul(q, kw) / c ** 0.5, dim=-1)
    return torch.matmul(prob, vw).view(b, h, w, flow.shape[1]).permute(0, 3, 1, 2).contiguous()


# ---- unimatch/matching.py -------------------------------------------------------------------------------------
def global_correlation_softmax_bidir(f0, f1, bidir=True):          # matching.py:10-39 (pred_bidir_flow: both directions stacked)
    b, c, h, w = f0.shape
    corr = torch.matmul(f0.view(b, c, -1).permute(0, 2, 1), f1.view(b, c, -1)) / c ** 0.5
    if bidir:
        corr = torch.cat((corr, corr.permute(0, 2, 1)), dim=0)
    nb = corr.shape[0]
    grid = coords_grid(nb, h, w, f0.dtype)
    prob = F.softmax(corr, dim=-1)
    corresp = torch.matmul(prob, grid.view(nb, 2, -1).permute(0, 2, 1)).view(nb, h, w, 2).permute(0, 3, 1, 2)
    return corresp - grid


def _window_grid(r, dtype):                                        # geometry.py:28-35 + matching.py:52-56
    lin = torch.linspace(-r, r, 2 * r + 1)
    x, y = torch.meshgrid([lin, lin], indexing="ij")
    return torch.stack((x, y), -1).transpose(0, 1).to(dtype).reshape(-1, 2)


def _normalize_coords(coords, h, w):                               # geometry.py:38-41
    c = torch.tensor([(w - 1) / 2.0, (h - 1) / 2.0], dtype=coords.dtype)
    return (coords - c) / c


def local_correlation_softmax(f0, f1, r):                          # matching.py:42-86
    b, c, h, w = f0.shape
    init = coords_grid(b, h, w, f0.dtype)
    coords = init.view(b, 2, -1).permute(0, 2, 1)
    sample = coords.unsqueeze(-2) + _window_grid(r, f0.dtype)[None, None]
    valid = (sample[..., 0] >= 0) & (sample[..., 0] < w) & (sample[..., 1] >= 0) & (sample[..., 1] < h)
    win = F.grid_sample(f1, _normalize_coords(sample, h, w), padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)
    corr = torch.matmul(f0.permute(0, 2, 3, 1).view(b, h * w, 1, c), win).view(b, h * w, -1) / c ** 0.5
    corr[~valid] = -1e9
    prob = F.softmax(corr, -1)
    corresp = torch.matmul(prob.unsqueeze(-2), sample).squeeze(-2).view(b, h, w, 2).permute(0, 3, 1, 2)
    return corresp - init


def local_correlation_with_flow(f0, f1, flow, r):                  # matching.py:89-126
    b, c, h, w = f0.shape
    coords = coords_grid(b, h, w, f0.dtype).view(b, 2, -1).permute(0, 2, 1)
    sample = coords.unsqueeze(-2) + _window_grid(r, f0.dtype)[None, None] + flow.view(b, 2, -1).permute(0, 2, 1).unsqueeze(-2)
    win = F.grid_sample(f1, _normalize_coords(sample, h, w), padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)
    corr = torch.matmul(f0.permute(0, 2, 3, 1).view(b, h * w, 1, c), win).view(b, h * w, -1) / c ** 0.5
    return corr.view(b, h, w, -1).permute(0, 3, 1, 2).contiguous()


# ---- unimatch/reg_refine.py -----------------------------------------------------------------------------------
def _c(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def basic_update_block(sd, net, inp, corr, flow, want_mask):       # reg_refine.py:58-122
    p = "refine."
    cor = F.relu(_c(sd, p + "encoder.convc1", corr, 0))
    cor = F.relu(_c(sd, p + "encoder.convc2", cor, 1))
    flo = F.relu(_c(sd, p + "encoder.convf1", flow, 3))
    flo = F.relu(_c(sd, p + "encoder.convf2", flo, 1))
    out = F.relu(_c(sd, p + "encoder.conv", torch.cat([cor, flo], dim=1), 1))
    x = torch.cat([inp, out, flow], dim=1)
    h = net
    for suf, pad in (("1", (0, 2)), ("2", (2, 0))):                # SepConvGRU, reg_refine.py:41-55
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(_c(sd, p + "gru.convz" + suf, hx, pad))
        r = torch.sigmoid(_c(sd, p + "gru.convr" + suf, hx, pad))
        q = torch.tanh(_c(sd, p + "gru.convq" + suf, torch.cat([r * h, x], dim=1), pad))
        h = (1 - z) * h + z * q
    delta = _c(sd, p + "flow_head.conv2", F.relu(_c(sd, p + "flow_head.conv1", h, 1)), 1)
    mask = _c(sd, p + "mask.2", F.relu(_c(sd, p + "mask.0", h, 1)), 0) if want_mask else None
    return h, mask, delta


# ---- unimatch/unimatch.py + __init__.py ---------------------------------------------------------------------------
def unimatch_flow_bidir(sd, img0, img1, num_reg_refine=6, dbg=None, bidir=True):   # unimatch.py:98-370 for the fixed configuration
    dtype = img0.dtype
    sd = {k: v.to(dtype) for k, v in sd.items()}
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=dtype).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=dtype).view(1, 3, 1, 1)
    img0, img1 = (img0 / 255.0 - mean) / std, (img1 / 255.0 - mean) / std          # utils.py:26-34
    feats = cnn_encoder(sd, torch.cat((img0, img1), dim=0))[::-1]                   # low res first (unimatch.py:67-82)
    f0l, f1l = [f.chunk(2, 0)[0] for f in feats], [f.chunk(2, 0)[1] for f in feats]
    if dbg is not None:
        dbg["feat_s0"], dbg["feat_s1"] = feats[0], feats[1]
    flow = None
    for scale, (splits, corr_r, prop_r) in enumerate(((2, -1, -1), (8, 4, 1))):
        f0, f1 = f0l[scale], f1l[scale]
        if scale > 0 and bidir:
            f0, f1 = torch.cat((f0, f1), dim=0), torch.cat((f1, f0), dim=0)         # unimatch.py:142-144
        f0_ori, f1_ori = f0, f1
        if scale > 0:
            flow = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2   # :157
            f1 = flow_warp(f1, flow)                                                # :171
        f0, f1 = feature_add_position(f0, f1, splits, 128)                          # :181
        f0, f1 = feature_transformer(sd, f0, f1, splits)                            # :184
        if dbg is not None:
            dbg["tf0_s%d" % scale] = f0
        if corr_r == -1:
            pred = global_correlation_softmax_bidir(f0, f1, bidir)                  # :208
        else:
            pred = local_correlation_softmax(f0, f1, corr_r)                        # :215
        flow = flow + pred if flow is not None else pred                            # :222
        if dbg is not None:
            dbg["flow_match_s%d" % scale] = flow
        if scale == 0 and bidir:
            f0 = torch.cat((f0, f1), dim=0)                                         # :237
        flow = self_attn_propagation(sd, f0, flow, prop_r)                          # :239-242
        if dbg is not None:
            dbg["flow_prop_s%d" % scale] = flow
        if scale == 1:
            for it in range(num_reg_refine):                                        # :276-357
                corr = local_correlation_with_flow(f0_ori, f1_ori, flow, 4)         # :311-316
                proj = F.conv2d(f0, sd["refine_proj.weight"], sd["refine_proj.bias"])   # :318
                net, inp = torch.chunk(proj, 2, dim=1)
                net, inp = torch.tanh(net), torch.relu(inp)
                last = it == num_reg_refine - 1
                net, up_mask, dflow = basic_update_block(sd, net, inp, corr, flow, want_mask=last)
                flow = flow + dflow                                                 # :331
                if dbg is not None:
                    dbg["flow_refine_%d" % it] = flow
            flow_up = upsample_flow_with_mask(flow, up_mask, 4)                     # :354
    return flow_up


def derive_matcher_inference_size(shape, max_area=500 * 900, padding_factor=32):   # methods/dmsct.py:58-74
    size = [int(np.ceil(shape[-2] / padding_factor)) * padding_factor, int(np.ceil(shape[-1] / padding_factor)) * padding_factor]
    aspect = shape[-1] / shape[-2]
    max_h = np.floor(np.sqrt(max_area / aspect))
    max_w = np.floor(max_h * aspect)
    max_size = [int(np.ceil(max_h / padding_factor)) * padding_factor, int(np.ceil(max_w / padding_factor)) * padding_factor]
    return max_size if size[0] * size[1] > max_size[0] * max_size[1] else size


def gmflow_forward(sd, img0, img1, inference_size, num_reg_refine=6, dbg=None, pred_bidir_flow=True, pred_bwd_flow=False):
    """unimatch/__init__.py:60-167: DMSCT's call (bidirectional + occlusion masks) by default; pred_bidir_flow=False is the
    one-direction form (:117-118 swaps the frames for pred_bwd_flow)"""
    transpose = img0.shape[-2] > img0.shape[-1]
    if transpose:
        img0, img1 = img0.transpose(-2, -1), img1.transpose(-2, -1)
    ori = img0.shape[-2:]
    resize = inference_size[0] != ori[0] or inference_size[1] != ori[1]
    if resize:
        img0 = F.interpolate(img0, size=list(inference_size), mode="bilinear", align_corners=True)
        img1 = F.interpolate(img1, size=list(inference_size), mode="bilinear", align_corners=True)
    if pred_bwd_flow:
        img0, img1 = img1, img0
    flow = unimatch_flow_bidir(sd, img0, img1, num_reg_refine, dbg, bidir=pred_bidir_flow)
    if resize:
        flow = F.interpolate(flow, size=list(ori), mode="bilinear", align_corners=True)
        flow = torch.stack([flow[:, 0] * ori[-1] / inference_size[-1], flow[:, 1] * ori[-2] / inference_size[-2]], dim=1)
    if transpose:
        flow = flow.transpose(-2, -1)
    if not pred_bidir_flow:
        return {"flow": flow}
    fwd, bwd = flow[::2], flow[1::2]
    fwd_occ, bwd_occ = forward_backward_consistency_check(fwd, bwd)
    return {"flow": fwd, "flow_bwd": bwd, "fwd_occ": fwd_occ.unsqueeze(1), "bwd_occ": bwd_occ.unsqueeze(1)}


# ---- methods/dmsct.py:76-114: the glue around the matcher (D1).  The smp encoder / decoder / head are third-party
# code absent offline ("parity unpinned"); this restates everything between them. -------------------
def dmsct_pad_size(shape, encoder_depth=4):                         # dmsct.py:76-82
    f = 2 ** encoder_depth
    return [0, (shape[-1] % f != 0) * (f - shape[-1] % f), 0, (shape[-2] % f != 0) * (f - shape[-2] % f)]


def dmsct_fuse_features(flow, fwd_occ, features_target, features_reference, pad_size):   # dmsct.py:99-114
    flow = F.pad(flow, pad_size, mode="replicate")
    occ = F.pad(fwd_occ, pad_size, mode="replicate")
    out = []
    for idx, (ft, fr) in enumerate(zip(features_target, features_reference)):
        s = 2.0 ** -idx
        fl = F.interpolate(flow, scale_factor=s, mode="bilinear", align_corners=True) * s      # unimatch.py:84-90
        out.append(torch.cat([ft, flow_warp(fr, fl), F.interpolate(1 - occ, mode="nearest", scale_factor=s)], dim=1))
    return out
