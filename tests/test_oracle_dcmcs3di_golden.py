"""Pin oracle/dcmcs3di.py (functional float64 restatement) against goldens captured from the
real reference module, and pin the product module's parameter tree (names, shapes, init order)."""
import os

import numpy as np
import torch

from oracle import dcmcs3di as odc
from tests.dcmcs3di_common import build_model, fingerprint


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_state_dict_names_shapes_and_init_match_reference(golden_dir):
    g = _g(golden_dir, "dcmcs3di_small.npz")
    m = build_model()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(s) for s in g["state_names"]]
    assert len(sd) == 114 and sum(v.numel() for v in sd.values()) == 1888323
    np.testing.assert_allclose(fingerprint(sd), g["state_fingerprint"], rtol=0, atol=0)
    g2 = _g(golden_dir, "dcmcs3di_shallow.npz")
    m2 = build_model(seed=3, extraction_layers=2, transfer_layers=1, channels=64)
    assert list(m2.state_dict().keys()) == [str(s) for s in g2["state_names"]]


def _check(g, name, res, tol=2e-5):
    t = torch.from_numpy
    for k in ("corrected", "pre_clamp", "warped_rgb", "colsum"):
        np.testing.assert_allclose(res[k].numpy(), g[name + "/" + k], rtol=3e-6, atol=tol, err_msg=k)
    for k in ("fea_left", "fea_right", "fea_warped"):
        np.testing.assert_allclose(res[k][:, ::8].numpy(), g[name + "/" + k + "_c8"], rtol=3e-6, atol=tol, err_msg=k)
    for k in ("att_r2l", "att_l2r", "cost_r2l", "cost_l2r"):
        np.testing.assert_allclose(res[k][:, ::8].numpy(), g[name + "/" + k + "_h8"], rtol=3e-6, atol=tol, err_msg=k)
    # boolean mask: compare away from the discontinuity
    colsum = g[name + "/colsum"]
    safe = np.abs(colsum - 0.1) > 1e-3
    assert (res["valid_left"][:, 0].numpy()[safe] == g[name + "/valid_left"][:, 0][safe]).all()
    del t


def test_oracle_vs_reference_small(golden_dir):
    g = _g(golden_dir, "dcmcs3di_small.npz")
    sd = build_model().state_dict()
    for name in ("a", "b"):
        res = odc.forward(sd, torch.from_numpy(g[name + "/left"]), torch.from_numpy(g[name + "/right"]))
        _check(g, name, res)


def test_oracle_vs_reference_shallow_batch2(golden_dir):
    g = _g(golden_dir, "dcmcs3di_shallow.npz")
    sd = build_model(seed=3, extraction_layers=2, transfer_layers=1, channels=64).state_dict()
    res = odc.forward(sd, torch.from_numpy(g["left"]), torch.from_numpy(g["right"]), extraction_layers=2,
                      transfer_layers=1)
    for k in ("corrected", "pre_clamp", "warped_rgb", "colsum"):
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=3e-6, atol=2e-5, err_msg=k)
