import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def port():
    import pyoracle
    pyoracle.build()
    return pyoracle.Oracle("port")


@pytest.fixture(scope="session")
def reference():
    import pyoracle
    pyoracle.build()
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/libdelly_ref.so not built (needs /root/reference)")
    return pyoracle.Oracle("reference")


@pytest.fixture(scope="session")
def gpu_ctx():
    from delly_amd import refine
    ctx = refine.Context()
    yield ctx
    ctx.close()
