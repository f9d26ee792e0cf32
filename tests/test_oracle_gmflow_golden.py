"""Pin oracle/gmflow.py (functional restatement of the GMFlow matcher in DMSCT's configuration) against goldens
captured from the real reference `unimatch.GMFlow` (tests/golden/make_golden_gmflow.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import gmflow as og
from tests.gmflow_common import procedural_state, test_pair as make_pair


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "gmflow_small.npz"), allow_pickle=False)


def state_from_fixture(g):
    shapes = [tuple(int(x) for x in s[:n]) for s, n in zip(g["state_shapes"], g["state_ndim"])]
    return procedural_state(g["state_names"], shapes)


@pytest.mark.parametrize("tag,hw,seed", [("a", (135, 240), 1), ("b", (96, 128), 2)])
def test_oracle_vs_reference(golden_dir, tag, hw, seed):
    g = _g(golden_dir)
    sd = state_from_fixture(g)
    assert len(sd) == 152 and sum(v.numel() for v in sd.values()) == 7360688
    img0, img1 = make_pair(seed, *hw)
    size = og.derive_matcher_inference_size((1, 3) + hw)
    assert list(size) == list(g[tag + "/size"])
    dbg = {}
    with torch.no_grad():
        res = og.gmflow_forward(sd, img0, img1, size, dbg=dbg)
    tol = dict(rtol=2e-4, atol=2e-3)          # float32 reference vs float32 restatement, different op grouping
    np.testing.assert_allclose(dbg["feat_s0"][:, ::16].numpy(), g[tag + "/feat_s0_c16"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dbg["feat_s1"][:, ::16].numpy(), g[tag + "/feat_s1_c16"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dbg["tf0_s0"][:, ::16].numpy(), g[tag + "/tf0_s0_c16"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dbg["flow_match_s0"].numpy(), g[tag + "/flow_match_s0"], **tol)
    np.testing.assert_allclose(dbg["flow_prop_s0"].numpy(), g[tag + "/flow_prop_s0"], **tol)
    np.testing.assert_allclose(dbg["tf0_s1"][:, ::16].numpy(), g[tag + "/tf0_s1_c16"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dbg["flow_match_s1"].numpy(), g[tag + "/flow_match_s1"], **tol)
    np.testing.assert_allclose(dbg["flow_prop_s1"].numpy(), g[tag + "/flow_prop_s1"], **tol)
    for i in range(6):
        np.testing.assert_allclose(dbg["flow_refine_%d" % i].numpy(), g[tag + "/flow_refine_%d" % i], rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(res["flow"].numpy(), g[tag + "/flow"], rtol=1e-3, atol=2e-2)
    np.testing.assert_allclose(res["flow_bwd"].numpy(), g[tag + "/flow_bwd"], rtol=1e-3, atol=2e-2)
    agree = (res["fwd_occ"].numpy() == g[tag + "/fwd_occ"]).mean()
    assert agree > 0.995


@pytest.mark.parametrize("tag,hw,seed", [("b", (96, 128), 2)])
def test_oracle_one_direction_vs_reference(golden_dir, tag, hw, seed):
    """pred_bidir_flow=False / pred_bwd_flow=True (unimatch/__init__.py:60-67) of the restatement against the real reference"""
    g = _g(golden_dir)
    u = np.load(os.path.join(golden_dir, "gmflow_uni.npz"), allow_pickle=False)
    sd = state_from_fixture(g)
    img0, img1 = make_pair(seed, *hw)
    size = og.derive_matcher_inference_size((1, 3) + hw)
    for name, kw in (("fwd", {}), ("bwd", {"pred_bwd_flow": True})):
        dbg = {}
        with torch.no_grad():
            res = og.gmflow_forward(sd, img0, img1, size, dbg=dbg, pred_bidir_flow=False, **kw)
        assert set(res.keys()) == {"flow"}
        np.testing.assert_allclose(dbg["flow_match_s0"].numpy(), u["%s/%s/flow_match_s0" % (tag, name)], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(dbg["flow_prop_s1"].numpy(), u["%s/%s/flow_prop_s1" % (tag, name)], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(res["flow"].numpy(), u["%s/%s/flow" % (tag, name)], rtol=1e-3, atol=2e-2)


def test_inference_size_rule():
    assert og.derive_matcher_inference_size((1, 3, 540, 960)) == [512, 896]
    assert og.derive_matcher_inference_size((1, 3, 135, 240)) == [160, 256]
