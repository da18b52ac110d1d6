"""C-ABI surface: the library loads and exports exactly what include/mlgnn.h declares; argument
errors come back as codes (no compute launched: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mlgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlgnn_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mlgnn import _lib
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert hasattr(_lib.lib, name), name


def test_version_and_error_codes():
    from mlgnn import _lib
    lib = _lib.lib
    assert lib.mlgnn_version() == 7
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 1, 0, 0) == 8 * 2 * 128      # 8 workgroups minimum
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 8, 0, 0) == 8 * 9 * 128
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 9, 0, 0) == -2
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(-1, 4, 0, 1, 0, 0) == -2
    # softmax: + flag (4) + rescaled cotangent (10*128 fp32 / bf16)
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 0, 3, 0) == 4 + 1280
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 1, 0, 3, 0) == 4 + 640
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 0, 3, 1) == 0
    null = [None] * 13
    # N = 0 is a no-op, bad dtype / mode / NULL pointers are reported, nothing is launched
    assert lib.mlgnn_csr_aggregate_fwd(*null, 0, 8, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == 0
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 7, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == -4
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 9, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == -3
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 1, 0, 0, 2, 1.0, 1.0, None, None, 1e-7, 0, None) == -3   # weighted+max
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == -1
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 2, 1, 3, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == -3   # edge rank 3
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 0, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None) == -2


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    from mlgnn import _lib
    monkeypatch.setattr(_lib, "LIB", str(tmp_path / "nope.so"))
    try:
        _lib._load()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a missing libmlgnn.so must raise")
    importlib.reload(_lib)


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from mlgnn import CSRGraph, gen_aggregate
    g = CSRGraph(torch.tensor([[0, 1], [1, 0]]), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gen_aggregate(torch.zeros(2, 4), g, None, aggr="add")
