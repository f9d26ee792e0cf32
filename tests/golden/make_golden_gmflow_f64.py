#!/usr/bin/env python3
"""float64 anchors for the GMFlow stages: what separates ROUNDING from ERROR in a 150-layer random-weight
network.  oracle/gmflow.py is run (a) in float32, where it must reproduce the reference fixtures gmflow_small.npz (made by
make_golden_gmflow.py from the real `unimatch.GMFlow`) -- asserted here, stage by stage, before anything is written -- and
(b) in float64 on the same state and pairs.  Stored per stage: the float64 values and d32 = max |reference float32 - float64|, the
distance the reference's OWN arithmetic keeps from exact.  tests/test_gmflow_gpu.py then asserts
    max |HIP - float64|  <=  1.5 x d32      (and prints the ratio)
next to the absolute STAGE_BOUNDS.  (The real reference cannot run in float64: geometry.py:17,33,40 and position.py:31-38 pin
float32.)  Runs in this container on CPU:   python3 -B tests/golden/make_golden_gmflow_f64.py     -> tests/golden/gmflow_f64.npz"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))                                   # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))                  # repo root (oracle/)
from gmflow_common import procedural_state, test_pair                      # noqa: E402
from oracle import gmflow as og                                            # noqa: E402

STAGES = ["feat_s0", "feat_s1", "tf0_s0", "flow_match_s0", "flow_prop_s0", "tf0_s1", "flow_match_s1", "flow_prop_s1"] + ["flow_refine_%d" % i for i in range(6)]
C16 = {"feat_s0", "feat_s1", "tf0_s0", "tf0_s1"}                           # the fixtures keep every 16th channel of these


def run(sd, img0, img1, size, dtype):
    dbg = {}
    with torch.no_grad():
        res = og.gmflow_forward(sd, img0.to(dtype), img1.to(dtype), size, dbg=dbg)
    out = {k: (dbg[k][:, ::16] if k in C16 else dbg[k]).double().numpy() for k in STAGES}
    out["flow"], out["flow_bwd"] = res["flow"].double().numpy(), res["flow_bwd"].double().numpy()
    return out


def main():
    g = np.load(os.path.join(OUT, "gmflow_small.npz"), allow_pickle=False)
    shapes = [tuple(int(x) for x in s[:n]) for s, n in zip(g["state_shapes"], g["state_ndim"])]
    sd = procedural_state([str(s) for s in g["state_names"]], shapes)
    fix = {}
    for tag, (h, w), seed in (("a", (135, 240), 1), ("b", (96, 128), 2)):
        img0, img1 = test_pair(seed, h, w)
        size = og.derive_matcher_inference_size((1, 3, h, w))
        r32, r64 = run(sd, img0, img1, size, torch.float32), run(sd, img0, img1, size, torch.float64)
        for k in r64:
            key = k + "_c16" if k in C16 else k
            ref = g[tag + "/" + key].astype(np.float64)
            pin = float(np.abs(r32[k] - ref).max())
            assert pin <= 1e-6 * max(1.0, float(np.abs(ref).max())), ("the float32 oracle left the reference fixture", tag, k, pin)
            fix["%s/%s" % (tag, key)] = r64[k].astype(np.float32)          # stored to float32: 1e-7 relative, far below every bound
            fix["%s/%s/d32" % (tag, key)] = np.float64(np.abs(ref - r64[k]).max())
            print("%s %-16s oracle32 vs reference %.1e   reference32 vs float64 %.3e" % (tag, key, pin, fix["%s/%s/d32" % (tag, key)]))
    np.savez_compressed(os.path.join(OUT, "gmflow_f64.npz"), torch=torch.__version__, **fix)


if __name__ == "__main__":
    main()
