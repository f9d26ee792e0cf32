"""Shared helpers for the DCMCS3DI tests: rebuild the golden state_dict from the seed recipe."""
import numpy as np
import torch


def build_model(seed=0, **kw):
    """Same recipe as tests/golden/make_golden_dcmcs3di.py:build, on the PRODUCT module:
    default init under torch.manual_seed, query/key weights x16, last bias 0.5."""
    from methods.dcmcs3di import DCMCS3DI
    torch.manual_seed(seed)
    m = DCMCS3DI(**kw).eval()
    with torch.no_grad():
        m.matcher.query.weight.mul_(16)
        m.matcher.key.weight.mul_(16)
        m.transfer[-1].bias.fill_(0.5)
    return m


def fingerprint(sd):
    return np.array([[float(v.double().sum()), float((v.double() ** 2).sum()), float(v.flatten()[0]),
                      float(v.flatten()[-1])] for v in sd.values()])
