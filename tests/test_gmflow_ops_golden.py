"""GMFlow primitives and the DMSCT glue against fixtures produced by the REFERENCE's own functions
(tests/golden/make_golden_gmflow_ops.py: unimatch/geometry.py:68-99, utils.py:137-155, matching.py:10-126,
attention.py:169-256, methods/dmsct.py:84-116) on inputs with MIXED occlusion masks.

* not-gpu tests: the oracle (oracle/gmflow.py) reproduces the fixtures -> it is pinned for these functions;
* gpu tests: the HIP kernels (through the C ABI) reproduce them, float32 rounding level; boolean masks must agree on
  every pixel whose test statistic is not within 1e-4 of its threshold.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import gmflow as og                      # noqa: E402
from tests.gmflow_common import procedural_tensor    # noqa: E402


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "gmflow_ops.npz"), allow_pickle=False)


def t(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def close(a, b, msg, atol, rtol=2e-6):      # the fixtures are float32 results of the reference: |x| * 2^-23 of rounding
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), np.asarray(b, np.float64), rtol=rtol, atol=atol, err_msg=msg)


def prop_state(dtype):
    names = ("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias")
    shapes = ((128, 128), (128,), (128, 128), (128,))
    return {"feature_flow_attn." + n: procedural_tensor("feature_flow_attn." + n, s).to(dtype) for n, s in zip(names, shapes)}


# ---- oracle vs the reference-run fixtures (CPU) -------------------------------------------------------------------------
def test_oracle_geometry(g):
    close(og.flow_warp(t(g["warp/feature"]), t(g["warp/flow"])), g["warp/out"], "flow_warp", 2e-6)
    close(og.upsample_flow_with_mask(t(g["up/flow"]), t(g["up/mask"]), 4), g["up/out"], "convex upsample", 2e-6)
    for tag in ("fb_a", "fb_b"):
        fo, bo = og.forward_backward_consistency_check(t(g[tag + "/fwd"], torch.float32), t(g[tag + "/bwd"], torch.float32))
        assert np.array_equal(fo.numpy(), g[tag + "/fwd_occ"]) and np.array_equal(bo.numpy(), g[tag + "/bwd_occ"])   # float32 like the reference: exact
        assert 0.1 < g[tag + "/fwd_occ"].mean() < 0.6            # the fixture really is a mixed mask
        fo, bo = og.forward_backward_consistency_check(t(g[tag + "/fwd"]), t(g[tag + "/bwd"]))
        for got, want, margin in ((fo, g[tag + "/fwd_occ"], g[tag + "/fwd_margin"]), (bo, g[tag + "/bwd_occ"], g[tag + "/bwd_margin"])):
            sure = np.abs(margin) > 1e-4
            assert sure.mean() > 0.99 and np.array_equal(got.numpy()[sure], want[sure])


def test_oracle_matching_and_propagation(g):
    f0, f1 = t(g["corr/f0"]), t(g["corr/f1"])
    close(og.global_correlation_softmax_bidir(f0, f1), g["corr/global_flow"], "global correlation", 2e-5)
    close(og.local_correlation_softmax(f0, f1, 4), g["corr/local_flow"], "local correlation", 2e-5)
    close(og.local_correlation_with_flow(f0, f1, t(g["corr/flow_in"]), 4), g["corr/with_flow"], "correlation with flow", 2e-5)
    sd = prop_state(torch.float64)
    close(og.self_attn_propagation(sd, t(g["prop/feature"]), t(g["prop/flow"]), -1), g["prop/global"], "global propagation", 2e-5)
    close(og.self_attn_propagation(sd, t(g["prop/feature"]), t(g["prop/flow"]), 1), g["prop/local_r1"], "3x3 propagation", 2e-5)


def test_oracle_dmsct_glue(g):
    chans = tuple(int(c) for c in g["enc_channels"])
    for key in g.files:
        if key.startswith("size/"):
            h, w = (int(v) for v in key[5:].split("x"))
            assert og.derive_matcher_inference_size((1, 3, h, w)) == [int(v) for v in g[key]]
    for tag in ("glue_a", "glue_b"):
        target = t(g[tag + "/target"])
        pad = og.dmsct_pad_size(target.shape)
        assert pad == [int(v) for v in g[tag + "/pad_size"]]
        assert og.derive_matcher_inference_size(target.shape) == [int(v) for v in g[tag + "/inference_size"]]
        fused = [g[tag + "/fused_%d" % i] for i in range(5)]
        # the fused tensors are [f_t | warped f_r | 1 - occ]: recover the encoder features the reference run used
        ft = [t(f[:, :c]) for f, c in zip(fused, chans)]
        fr = _stub_encoder(torch.nn.functional.pad(t(g[tag + "/reference"], torch.float32), pad, mode="replicate"), chans)
        out = og.dmsct_fuse_features(t(g[tag + "/flow"]), t(g[tag + "/fwd_occ"]), ft, [f.double() for f in fr], pad)
        for i, (a, b) in enumerate(zip(out, fused)):
            close(a, b, "%s scale %d" % (tag, i), 2e-5)
            vis = b[:, -1]
            assert 0.2 < (vis == 0).mean() < 0.5 and set(np.unique(vis)) == {0.0, 1.0}       # the occlusion channel is mixed


def _stub_encoder(x, chans):
    """the feature pyramid of make_golden_gmflow_ops.py's StubEncoder (test data generator, not reference code)"""
    feats = [x]
    for i in range(1, len(chans)):
        gen = torch.Generator().manual_seed(100 + i)
        mix = torch.randn(chans[i], 3, generator=gen)
        feats.append(torch.sin(torch.einsum("oc,bchw->bohw", mix.to(x.dtype), torch.nn.functional.avg_pool2d(x, 2 ** i)) * 3.0))
    return feats


# ---- HIP kernels vs the reference-run fixtures (GPU) ----------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


def d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().cuda()


def tok(x):
    return x.flatten(2).transpose(1, 2).contiguous()


@pytest.mark.gpu
def test_hip_geometry(g, hip):
    close(hip.flow_warp(d(g["warp/feature"]), d(g["warp/flow"])), g["warp/out"], "flow_warp", 2e-5)
    close(hip.convex_upsample(d(g["up/flow"]), d(g["up/mask"]), 4), g["up/out"], "convex upsample", 2e-5)
    for tag in ("fb_a", "fb_b"):
        fo, bo = hip.fb_check(d(g[tag + "/fwd"]), d(g[tag + "/bwd"]))
        for got, want, margin in ((fo, g[tag + "/fwd_occ"], g[tag + "/fwd_margin"]), (bo, g[tag + "/bwd_occ"], g[tag + "/bwd_margin"])):
            got = got.cpu().numpy()
            sure = np.abs(margin) > 1e-4
            assert got.shape == want.shape and set(np.unique(got)) <= {0.0, 1.0}
            assert sure.mean() > 0.99 and np.array_equal(got[sure], want[sure]), tag
            assert 0.1 < got.mean() < 0.6


@pytest.mark.gpu
def test_hip_matching_and_propagation(g, hip):
    f0, f1 = d(g["corr/f0"]), d(g["corr/f1"])
    b, c, h, w = f0.shape
    t0, t1 = tok(f0), tok(f1)
    # global correlation softmax, both directions (matching.py:10-39): attention with the pixel grid as the value
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack([xx, yy], dim=-1).float().reshape(1, h * w, 2).repeat(b, 1, 1).contiguous().cuda()
    corresp = torch.cat((hip.attention_tokens(t0, t1, grid), hip.attention_tokens(t1, t0, grid)), dim=0)
    flow = (corresp - torch.cat((grid, grid), dim=0)).transpose(1, 2).reshape(2 * b, 2, h, w)
    close(flow, g["corr/global_flow"], "global correlation", 2e-4)
    close(hip.local_corr_softmax(t0, t1, h, w, 4), g["corr/local_flow"], "local correlation", 2e-4)
    close(hip.local_corr_flow(t0, t1, d(g["corr/flow_in"]), 4), g["corr/with_flow"], "correlation with flow", 2e-4)
    sd = {k: v.cuda() for k, v in prop_state(torch.float32).items()}
    ft, pflow = tok(d(g["prop/feature"])), d(g["prop/flow"])
    q = hip.linear_tokens(ft, sd["feature_flow_attn.q_proj.weight"], sd["feature_flow_attn.q_proj.bias"])
    kq = hip.linear_tokens(q, sd["feature_flow_attn.k_proj.weight"], sd["feature_flow_attn.k_proj.bias"])       # k_proj(q_proj(x))
    out = hip.attention_tokens(q, kq, tok(pflow)).transpose(1, 2).reshape(b, 2, h, w)
    close(out, g["prop/global"], "global propagation", 2e-4)
    k = hip.linear_tokens(ft, sd["feature_flow_attn.k_proj.weight"], sd["feature_flow_attn.k_proj.bias"])
    close(hip.local_attn_prop(q, k, pflow, 1), g["prop/local_r1"], "3x3 propagation", 2e-4)


@pytest.mark.gpu
def test_hip_dmsct_glue_and_forward(g, hip):
    """`DMSCT.fuse_features` and the whole `DMSCT.forward` (methods/dmsct.py:84-116) with the stand-in matcher and stub
    encoder / decoder / head of the golden run, against what the reference's forward produced."""
    from methods.dmsct import DMSCT
    chans = tuple(int(c) for c in g["enc_channels"])

    class Enc(torch.nn.Module):
        def forward(self, x):
            return [f.to(x.device) for f in _stub_encoder(x.cpu(), chans)]

    class Dec(torch.nn.Module):
        def forward(self, *features):
            self.seen = features
            return features[0][:, :3]

    class Head(torch.nn.Module):
        def forward(self, x):
            return 0.25 * x - 0.05

    for tag in ("glue_a", "glue_b"):
        model = DMSCT(encoder=Enc(), decoder=Dec(), head=Head()).cuda()
        flow, occ = d(g[tag + "/flow"]), d(g[tag + "/fwd_occ"])
        seen = {}

        def fake_match(target, reference, _flow=flow, _occ=occ, _seen=seen):
            _seen["size"] = DMSCT.derive_matcher_inference_size(reference.shape)
            return {"flow": _flow, "fwd_occ": _occ}

        model.match = fake_match
        target, reference = d(g[tag + "/target"]), d(g[tag + "/reference"])
        out = model(target, reference)
        assert seen["size"] == [int(v) for v in g[tag + "/inference_size"]]
        assert model.derive_pad_size(reference.shape) == [int(v) for v in g[tag + "/pad_size"]]
        for i, f in enumerate(model.decoder.seen):
            close(f, g[tag + "/fused_%d" % i], "%s scale %d" % (tag, i), 1e-4)
        close(out, g[tag + "/out"], tag + " output", 1e-4)
