import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build what is missing.
    (Only when missing: a stale-by-mtime rebuild on the GPU box would burn GPU time.)"""
    from tantivy_amd import build as product_build

    if not os.path.exists(product_build.LIB):
        product_build.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        from oracle import oracle as O

        O.build()


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
