import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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


@pytest.fixture(params=['bf16x3', 'f32'])
def both_gemm_precisions(request, monkeypatch):
    """Parity tests run under BOTH precisions of the activation x weight products with the SAME tolerances: 'bf16x3' (the default:
    fp32-class split-bf16 products where a kernel takes the shape) and 'f32' (the exact fp32 MFMA everywhere).  Use through
    `pytestmark = pytest.mark.usefixtures('both_gemm_precisions')` or on single tests."""
    from geographconv_amd import ops
    monkeypatch.setattr(ops, 'GEMM_PRECISION', request.param)
    return request.param
