#!/usr/bin/env python3
"""Generate golden vectors for the GMFlow matcher exactly as DMSCT calls it (methods/dmsct.py:85-94) by running
the REAL reference `unimatch.GMFlow` on CPU (build container only):

    python3 -B tests/golden/make_golden_gmflow.py

`torch.hub.load_state_dict_from_url` is patched (the constructor otherwise downloads weights,
unimatch/__init__.py:55); parameters are then overwritten by the procedural, name-derived state of
tests/gmflow_common.py.  Intermediates are captured with forward hooks; the reference source is untouched.
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))           # tests/
sys.path.insert(0, "/root/reference")
torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": {}}
from unimatch import GMFlow  # noqa: E402
from gmflow_common import procedural_state, test_pair  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from oracle.gmflow import derive_matcher_inference_size  # noqa: E402  (same arithmetic as DMSCT's static method)


def run(model, img0, img1, size):
    cap = {"tf": [], "prop_in": [], "prop_out": [], "refine_in": [], "refine_out": []}
    hooks = [
        model.backbone.register_forward_hook(lambda m, i, o: cap.__setitem__("feats", [t.detach().clone() for t in o])),
        model.transformer.register_forward_hook(lambda m, i, o: cap["tf"].append(o[0].detach().clone())),
        model.feature_flow_attn.register_forward_pre_hook(lambda m, i: cap["prop_in"].append(i[1].detach().clone())),
        model.feature_flow_attn.register_forward_hook(lambda m, i, o: cap["prop_out"].append(o.detach().clone())),
        model.refine.register_forward_pre_hook(lambda m, i: cap["refine_in"].append(i[3].detach().clone())),
        model.refine.register_forward_hook(lambda m, i, o: cap["refine_out"].append(o[2].detach().clone())),
    ]
    with torch.no_grad():
        res = model(img0, img1, inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True)
    for h in hooks:
        h.remove()
    out = {k: v.numpy() for k, v in res.items()}
    out["feat_s1_c16"] = cap["feats"][0][:, ::16].numpy()        # backbone output list is high -> low resolution
    out["feat_s0_c16"] = cap["feats"][1][:, ::16].numpy()
    out["tf0_s0_c16"], out["tf0_s1_c16"] = cap["tf"][0][:, ::16].numpy(), cap["tf"][1][:, ::16].numpy()
    out["flow_match_s0"], out["flow_match_s1"] = cap["prop_in"][0].numpy(), cap["prop_in"][1].numpy()
    out["flow_prop_s0"], out["flow_prop_s1"] = cap["prop_out"][0].numpy(), cap["prop_out"][1].numpy()
    for i, (fin, d) in enumerate(zip(cap["refine_in"], cap["refine_out"])):
        out["flow_refine_%d" % i] = (fin + d).numpy()
    return out


def main():
    model = GMFlow("mixdata").eval()
    names = list(model.state_dict().keys())
    shapes = [tuple(v.shape) for v in model.state_dict().values()]
    model.load_state_dict(procedural_state(names, shapes), strict=True)
    fix = {"state_names": np.array(names), "state_shapes": np.array([s + (0,) * (4 - len(s)) for s in shapes]),
           "state_ndim": np.array([len(s) for s in shapes])}
    for tag, (h, w), seed in (("a", (135, 240), 1), ("b", (96, 128), 2)):
        img0, img1 = test_pair(seed, h, w)
        size = derive_matcher_inference_size((1, 3, h, w))
        res = run(model, img0, img1, size)
        fix[tag + "/size"] = np.array(size)
        for k, v in res.items():
            fix[tag + "/" + k] = v
        print(tag, "inference size", size, "|flow| mean %.3f max %.3f" % (np.abs(res["flow"]).mean(), np.abs(res["flow"]).max()),
              "occ frac %.3f" % res["fwd_occ"].mean())
    np.savez_compressed(os.path.join(OUT, "gmflow_small.npz"), torch=torch.__version__, **fix)
    # the one-direction call forms of the same wrapper (unimatch/__init__.py:60-67: pred_bidir_flow=False, pred_bwd_flow), same
    # state and pairs; only the stages that differ from the bidirectional run are kept (gmflow_uni.npz)
    uni = {}
    for tag, (h, w), seed in (("a", (135, 240), 1), ("b", (96, 128), 2)):
        img0, img1 = test_pair(seed, h, w)
        size = derive_matcher_inference_size((1, 3, h, w))
        for name, kw in (("fwd", {}), ("bwd", {"pred_bwd_flow": True})):
            cap = {"prop_in": [], "prop_out": []}
            hooks = [model.feature_flow_attn.register_forward_pre_hook(lambda m, i: cap["prop_in"].append(i[1].detach().clone())),
                     model.feature_flow_attn.register_forward_hook(lambda m, i, o: cap["prop_out"].append(o.detach().clone()))]
            with torch.no_grad():
                res = model(img0, img1, inference_size=size, pred_bidir_flow=False, **kw)
            for hk in hooks:
                hk.remove()
            assert set(res.keys()) == {"flow"}
            uni["%s/%s/flow" % (tag, name)] = res["flow"].numpy()
            uni["%s/%s/flow_match_s0" % (tag, name)] = cap["prop_in"][0].numpy()
            uni["%s/%s/flow_prop_s1" % (tag, name)] = cap["prop_out"][1].numpy()
            print(tag, name, "one direction: flow", res["flow"].shape, "|flow| mean %.3f" % np.abs(res["flow"]).mean())
    np.savez_compressed(os.path.join(OUT, "gmflow_uni.npz"), torch=torch.__version__, **uni)
    print("wrote gmflow goldens:", sum(int(np.prod(s)) for s in shapes), "parameters,", len(names), "tensors")


if __name__ == "__main__":
    main()
