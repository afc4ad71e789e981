import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU (or without the built library) skips the gpu-marked tests instead of
    failing them; on a GPU box a missing library is still an error (the product path must fail loudly)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X: torch.cuda.is_available() is False")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
