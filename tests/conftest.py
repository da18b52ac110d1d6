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


@pytest.fixture(autouse=True)
def _canary_guard(request):
    """MLGNN_CANARY=1 (debug run of the suite under the guard-band allocator, csrc/canary.hip): the bands are compared
    after every C-ABI call by the binding; once more at the end of each test, which also covers ATen's kernels."""
    yield
    if os.environ.get("MLGNN_CANARY") == "1":
        import torch
        if torch.cuda.is_available():
            from mlgnn import _lib
            _lib.canary_check("test " + request.node.nodeid)


def pytest_sessionfinish(session, exitstatus):
    if os.environ.get("MLGNN_CANARY") == "1":
        import json
        import torch
        if torch.cuda.is_available():
            from mlgnn import _lib
            stats = dict(_lib.canary_stats(), exitstatus=int(exitstatus), tests=session.testscollected)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "canary_suite.json"), "w") as fh:
                json.dump(stats, fh)
            print("\nMLGNN_CANARY:", json.dumps(stats))
