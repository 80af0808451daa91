import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; if someone runs them on a box without a GPU, fail loudly
    # instead of silently passing.
    pass


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_present():
        pytest.fail("this test needs a GPU (torch.cuda.is_available() is False)")
    import torch

    torch.cuda.set_device(0)
    return torch.device("cuda:0")
