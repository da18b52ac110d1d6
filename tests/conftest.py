"""pytest configuration: ``gpu`` marker, import roots, fixture loader."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multilevel-gnn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _ensure_native():
    """Host-logic tests import ``mlgnn``, which dlopens libmlgnn.so: build it when missing/stale
    (hipcc cross-compiles gfx950 without a GPU).  On the GPU box the prebuilt .so travels along."""
    import build_native
    try:
        build_native.build(force=False, verbose=False)
    except Exception as exc:                      # no hipcc: only acceptable if a library exists
        if not os.path.exists(build_native.LIB):
            raise RuntimeError("libmlgnn.so missing and cannot be built: %s" % exc)


_ensure_native()
