"""pytest configuration: the `gpu` marker and import paths."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; when a GPU is absent they skip loudly.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _tower_mode_from_env():
    """DFM_TEST_TOWER_MODE=1: run the whole GPU suite with the tower's backward GEMMs on the bf16 x 3 path
    (dfm_tower_set_mode) — the acceptance test of that mode is that every parity test passes with no tolerance edited."""
    mode = os.environ.get("DFM_TEST_TOWER_MODE")
    if mode is not None:
        import torch
        if torch.cuda.is_available():
            from deepfm_amd import _lib
            _lib.check(_lib.load().dfm_tower_set_mode(int(mode)))
    yield
