"""pytest configuration: markers and import paths.

* ``gpu`` marker = needs a real MI355X (run by the driver with ``-m gpu``).
* The drop-in tree ``color-transfer_amd/`` mirrors the reference's repo root
  (``methods/``, ``utils/``, ``configs/``), so it goes on ``sys.path`` exactly like
  the reference's root would.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "color-transfer_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP kernels through the C ABI)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["split", "split-ws", "exact"])
def conv_mode(request):
    """The arithmetic paths of the stride-1 convolutions: split MFMA (default; the 3x3 convolutions with 32 < cin <= 64 as Winograd
    F(2x2,3x3), csrc/conv_wino.hip), the same with those convolutions on the weight-stationary direct kernel (csrc/conv_ws.hip;
    "split-ws", yields "split"), and exact-f32 MFMA."""
    import ct_hip
    mode = "exact" if request.param == "exact" else "split"
    ct_hip.set_conv_mode(mode)
    ct_hip.set_conv_wino(request.param == "split")
    yield mode
    ct_hip.set_conv_mode("split")
    ct_hip.set_conv_wino(True)
