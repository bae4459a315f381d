import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure)."""
    import oracle

    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """rlhip context on cuda:0.  GPU tests FAIL (not skip) when the HIP extension cannot be used."""
    import torch

    assert torch.cuda.is_available(), "a -m gpu test was collected on a machine without a HIP device"
    from randlapack_amd.device import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _reset_ctx_options(request):
    """Options a GPU test set on the shared context (Context.set_option) go back to their defaults when the test ends."""
    yield
    if request.node.get_closest_marker("gpu") and "ctx" in request.fixturenames:
        c = request.getfixturevalue("ctx")
        for name in c.OPT:
            c.set_option(name, -1)
