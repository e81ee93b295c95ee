import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One smt_ctx on cuda:0.  torch is imported FIRST so that libsemtools_hip.so binds to the
    same libamdhip64 (SONAME libamdhip64.so.7) that torch already loaded."""
    import torch  # noqa: F401

    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback to test"
    import semtools_amd as smt

    ctx = smt.Context(0)
    yield ctx
    ctx.close()
