#!/opt/conda/bin/python3.9 -B
"""Golden vectors for the regrain half of methods/iterative.py by running the REAL reference and the real scikit-image
(0.18.3 in /opt/conda; build container only):

    /opt/conda/bin/python3.9 -B tests/golden/make_golden_regrain.py

Captured: skimage.transform.resize down / up (the third-party call the reference makes at iterative.py:68-71), `_solve`,
`_regrain` (float64 inputs, and the float32-target / float64-colour mix automated_color_grading produces from a Runner
frame), and `automated_color_grading` end to end with the rotations recorded (scipy's RNG stream is version dependent:
treated as data).  Only data is written."""
import importlib.util
import os
import warnings

import numpy as np
import scipy.stats
from skimage.transform import resize
import skimage

OUT = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_iterative", "/root/reference/methods/iterative.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def image(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([0.5 + 0.4 * np.sin(xx / 9.0 + seed) * np.cos(yy / 7.0), 0.15 + 0.7 * xx / w, 0.5 + 0.4 * np.cos((xx + 2 * yy) / 11.0)], -1)
    base[h // 3: h // 2, w // 4: w // 2] *= 0.5                      # an edge: psi / phi switch regimes there
    return np.clip(base + 0.03 * rng.standard_normal(base.shape), 0, 1)


def main():
    fix = {"skimage": skimage.__version__, "scipy": scipy.__version__, "numpy": np.__version__}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, (h, w) in {"a": (90, 120), "b": (101, 83), "c": (43, 47)}.items():
            tgt = image(3 + h, h, w)
            col = np.clip(tgt ** 0.8 * np.array([1.05, 0.9, 0.95]) + 0.04 * np.random.default_rng(h).standard_normal(tgt.shape), 0, 1)
            fix[tag + "/in"], fix[tag + "/col"] = tgt, col
            h2, w2 = (h + 1) // 2, (w + 1) // 2
            down = resize(tgt, (h2, w2))
            fix[tag + "/resize_down"], fix[tag + "/resize_up"] = down, resize(down, (h, w))
            fix[tag + "/solve_l0_n4"] = ref._solve(tgt, tgt, col, 4, 0)
            fix[tag + "/solve_l2_n7"] = ref._solve(col * 0.5 + tgt * 0.5, tgt, col, 7, 2)
            fix[tag + "/regrain"] = ref._regrain(tgt, col)
            fix[tag + "/regrain_f32in"] = ref._regrain(tgt.astype(np.float32), col)
            print(tag, (h, w), "regrain dtype", fix[tag + "/regrain"].dtype, fix[tag + "/regrain_f32in"].dtype,
                  "max |f32in - f64in| %.2e" % np.abs(fix[tag + "/regrain"] - fix[tag + "/regrain_f32in"]).max())
        # end to end with recorded rotations
        rots = []
        orig = scipy.stats.special_ortho_group.rvs

        def rvs(*a, **k):
            r = orig(*a, **k)
            rots.append(r.copy())
            return r
        scipy.stats.special_ortho_group.rvs = rvs
        np.random.seed(5)
        t32 = image(50, 64, 96).astype(np.float32)
        r32 = image(51, 48, 80).astype(np.float32)[:, ::-1].copy() * np.float32(0.8) + np.float32(0.1)
        fix["acg/target"], fix["acg/reference"] = t32, r32
        fix["acg/out"] = ref.automated_color_grading(t32, r32)
        fix["acg/rotations"] = np.stack(rots)
        scipy.stats.special_ortho_group.rvs = orig
        print("automated_color_grading:", fix["acg/out"].dtype, fix["acg/out"].shape, "rotations", fix["acg/rotations"].shape)
    np.savez_compressed(os.path.join(OUT, "regrain.npz"), **fix)


if __name__ == "__main__":
    main()
