import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


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
def big_map():
    """BASELINE config C2 map: 400x400 @ 0.04 m Perlin terrain + obstacles."""
    from synthetic import make_map
    return make_map(400, 0.04, seed=1234)


@pytest.fixture(scope="session")
def ctx_yaml():
    from art_planner_amd.context import Context
    c = Context(0, "yaml")
    yield c
    c.close()
