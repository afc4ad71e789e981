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


class _Forms:
    """Kernel-form switches for the parity tests: `form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")` hands the switch to every loaded build of
    the library through dyf_debug_set_form (include/dyffusion_hip_testing.h) -- the library reads no such switch from the environment.
    (The method names are monkeypatch's: these tests set environment variables until round 5.)"""

    def setenv(self, key, value):
        from dyffusion_amd import _lib
        _lib.set_form(key, str(value))

    def delenv(self, key, raising=False):
        from dyffusion_amd import _lib
        _lib.set_form(key, None)

    def env(self, **switches):
        """The environment of a child script that applies `switches` with tests.helpers.apply_test_forms()."""
        from dyffusion_amd import _lib
        merged = dict(_lib.forms(), **{k: str(v) for k, v in switches.items()})
        return dict(os.environ, DYF_TEST_FORMS=";".join(f"{k}={v}" for k, v in merged.items()))


@pytest.fixture
def form_switch():
    yield _Forms()
    from dyffusion_amd import _lib
    _lib.set_form(None)  # every switch back to the production policy
