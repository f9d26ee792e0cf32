"""GPU parity of the GMFlow matcher (DMSCT's configuration) against goldens captured from the real reference and
against the oracle; float32 rounding-level tolerances (the graph is ~150 layers deep)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.gmflow_common import procedural_state, test_pair as make_pair   # noqa: E402


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "gmflow_small.npz"), allow_pickle=False)


def build(g):
    from unimatch import GMFlow
    shapes = [tuple(int(x) for x in s[:n]) for s, n in zip(g["state_shapes"], g["state_ndim"])]
    m = GMFlow()
    assert list(m.state_dict().keys()) == [str(s) for s in g["state_names"]]         # same parameter tree as the reference
    assert [tuple(v.shape) for v in m.state_dict().values()] == shapes
    m.load_state_dict(procedural_state(g["state_names"], shapes), strict=True)
    return m.cuda()


@pytest.mark.parametrize("tag,hw,seed", [("a", (135, 240), 1), ("b", (96, 128), 2)])
def test_gmflow_vs_reference(golden_dir, tag, hw, seed):
    from oracle.gmflow import derive_matcher_inference_size
    g = _g(golden_dir)
    m = build(g)
    img0, img1 = make_pair(seed, *hw)
    size = derive_matcher_inference_size((1, 3) + hw)
    dbg = {}
    res = m(img0.cuda(), img1.cuda(), inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True, dbg=dbg)

    def chk(a, b, msg, rtol, atol):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=rtol, atol=atol, err_msg=msg)
    # tolerances = a few times the float32-vs-float64 spread of this graph (tests/gmflow_common.py), i.e. rounding level
    chk(dbg["feat_s0"][:, ::16], g[tag + "/feat_s0_c16"], "backbone 1/8", 2e-4, 2e-4)
    chk(dbg["feat_s1"][:, ::16], g[tag + "/feat_s1_c16"], "backbone 1/4", 2e-4, 2e-4)
    chk(dbg["tf0_s0"][:, ::16], g[tag + "/tf0_s0_c16"], "transformer s0", 1e-3, 1e-3)
    chk(dbg["flow_match_s0"], g[tag + "/flow_match_s0"], "global match", 1e-3, 1e-2)
    chk(dbg["flow_prop_s0"], g[tag + "/flow_prop_s0"], "propagation s0", 1e-3, 1e-2)
    chk(dbg["tf0_s1"][:, ::16], g[tag + "/tf0_s1_c16"], "transformer s1", 2e-3, 5e-3)
    chk(dbg["flow_match_s1"], g[tag + "/flow_match_s1"], "local match", 1e-3, 1e-1)
    chk(dbg["flow_prop_s1"], g[tag + "/flow_prop_s1"], "propagation s1", 1e-3, 1e-1)
    for i in range(6):
        chk(dbg["flow_refine_%d" % i], g[tag + "/flow_refine_%d" % i], "refine %d" % i, 2e-3, 1e-1)
    chk(res["flow"], g[tag + "/flow"], "flow", 2e-3, 3e-1)
    chk(res["flow_bwd"], g[tag + "/flow_bwd"], "flow_bwd", 2e-3, 3e-1)
    assert res["fwd_occ"].shape == (1, 1) + hw
    assert (res["fwd_occ"].cpu().numpy() == g[tag + "/fwd_occ"]).mean() > 0.99


def test_gmflow_rejects_other_configurations(golden_dir):
    from unimatch import GMFlow
    m = GMFlow().cuda()
    x = torch.rand(1, 3, 64, 64).cuda()
    with pytest.raises(NotImplementedError):
        m(x, x)                                    # pred_bidir_flow=False is not DMSCT's call
