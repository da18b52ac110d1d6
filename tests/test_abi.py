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
    assert lib.mlgnn_version() == 19
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 1, 0, 0) == (8 + 256) * 2 * 128      # 8 workgroups minimum + the long-row launch
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 8, 0, 0) == (8 + 256) * 9 * 128
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 9, 0, 0) == -2
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(-1, 4, 0, 1, 0, 0) == -2
    # softmax: + flag (4) + rescaled cotangent (10*128 fp32 / bf16)
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 0, 3, 0) == 4 + 1280
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 1, 0, 3, 0) == 4 + 640
    assert lib.mlgnn_csr_aggregate_bwd_workspace_floats(10, 128, 0, 0, 3, 1) == 0
    null = [None] * 13
    # N = 0 is a no-op, bad dtype / mode / NULL pointers are reported, nothing is launched
    assert lib.mlgnn_csr_aggregate_fwd(*null, 0, 8, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == 0
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 7, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -4
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 9, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -3
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 1, 0, 0, 2, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -3   # weighted+max
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -1
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 8, 0, 2, 1, 3, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -3   # edge rank 3
    assert lib.mlgnn_csr_aggregate_fwd(*null, 4, 0, 0, 2, 0, 0, 3, 1.0, 1.0, None, None, 1e-7, 0, None, None) == -2


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


def test_argument_errors_of_the_dense_and_embedding_entry_points():
    """Shape / dtype / mode / NULL errors are reported before anything is launched (no GPU needed): the bf16 variants
    of the tall GEMM, weight gradient and LayerNorm, and the embedding gradient."""
    from mlgnn import _lib
    lib = _lib.lib
    F32, BF16 = 0, 1
    # tall GEMM: supported shapes per storage type, workspace sizes
    assert lib.mlgnn_tallgemm_supported(1000, 128, 256, F32) == 1
    assert lib.mlgnn_tallgemm_supported(1000, 512, 256, F32) == 0          # fp32 image beyond LDS
    assert lib.mlgnn_tallgemm_supported(1000, 512, 256, BF16) == 1         # bf16: column slices
    assert lib.mlgnn_tallgemm_supported(1000, 40, 64, BF16) == 0 and lib.mlgnn_tallgemm_supported(1000, 64, 48, BF16) == 0
    assert lib.mlgnn_tallgemm_supported(1000, 64, 64, 5) == 0
    assert lib.mlgnn_tallgemm_workspace_bytes(256, 512, BF16) == 256 * 512 * 2
    assert lib.mlgnn_tallgemm_workspace_bytes(128, 256, F32) == 128 * 256 * 4 + 64
    tg = lambda dtype, ln, N=64, R=64, J=64, tr=0: lib.mlgnn_tallgemm_nt(None, None, tr, None, None, None, ln, None, None,
                                                                          0.0, None, None, None, None, 1 << 20, N, R, J,
                                                                          dtype, None)
    assert tg(BF16, 1) == -3                                               # LayerNorm modes are fp32-only
    assert tg(BF16, 0) == -1 and tg(F32, 0) == -1                          # NULL operands
    assert tg(BF16, 0, R=40) == -2 and tg(7, 0) == -4
    assert tg(BF16, 0, N=0) == 0 and tg(BF16, 0, tr=1) == -3             # transposed operand: fp32 only
    # weight gradient
    assert lib.mlgnn_linear_wgrad_workspace_floats(10000, 512, 256, BF16) > 0
    assert lib.mlgnn_linear_wgrad_workspace_floats(10000, 512, 200, BF16) == -2       # K % 128 != 0
    assert lib.mlgnn_linear_wgrad_workspace_floats(10000, 100, 256, BF16) == -2       # M % 64 != 0
    assert lib.mlgnn_linear_wgrad_workspace_floats(10000, 256, 256, F32) == -2        # fp32: 64 tiles > 32
    assert lib.mlgnn_linear_wgrad_workspace_floats(10000, 128, 128, 9) == -4
    wg = lambda dtype, M, K, ws=1 << 30: lib.mlgnn_linear_wgrad(None, None, None, None, None, None, None, None, ws, 1000, M, K, dtype, None)
    assert wg(BF16, 512, 256) == -1 and wg(BF16, 512, 200) == -2 and wg(3, 128, 128) == -4
    # LayerNorm: widths per storage type
    assert lib.mlgnn_layernorm_bwd_workspace_floats(1000, 512, F32) > 0 and lib.mlgnn_layernorm_bwd_workspace_floats(1000, 512, BF16) > 0
    assert lib.mlgnn_layernorm_bwd_workspace_floats(1000, 260, F32) == -2             # beyond 256 needs d % 8 == 0
    assert lib.mlgnn_layernorm_bwd_workspace_floats(1000, 516, BF16) == -2 and lib.mlgnn_layernorm_bwd_workspace_floats(1000, 520, F32) == -2
    ln = lambda dtype, d: lib.mlgnn_layernorm_act_fwd(None, None, None, None, None, None, None, None, 1.0, 10, d, 1e-5, 1,
                                                      dtype, None)
    assert ln(F32, 128) == -1 and ln(BF16, 100) == -2 and ln(4, 128) == -4
    # embedding gradient
    emb = lambda T, d, dtype=F32: lib.mlgnn_embedding_bwd(None, None, None, None, T, d, dtype, None)
    assert emb(0, 128) == 0 and emb(10, 128) == -1 and emb(10, 130) == -2 and emb(10, 128, BF16) == -4


def test_argument_errors_of_the_one_pass_linear_backward():
    """mlgnn_linear_bwd / mlgnn_tallgemm_bf16_shift: shapes, epilogues, NULL operands and workspace sizes are checked
    before anything is launched (no GPU needed)."""
    from mlgnn import _lib
    lib = _lib.lib
    LN, PLAIN, SHIFT = 0, 1, 2
    assert lib.mlgnn_linear_bwd_supported(640000, 128, 256, LN) == 1
    assert lib.mlgnn_linear_bwd_supported(640000, 256, 128, SHIFT) == 1 and lib.mlgnn_linear_bwd_supported(640000, 256, 128, PLAIN) == 1
    assert lib.mlgnn_linear_bwd_supported(640000, 256, 128, LN) == 0            # the LayerNorm epilogue: 128 -> 256 only
    assert lib.mlgnn_linear_bwd_supported(640000, 128, 128, PLAIN) == 0 and lib.mlgnn_linear_bwd_supported(0, 128, 256, LN) == 0
    assert lib.mlgnn_linear_bwd_supported(640000, 128, 256, 7) == 0
    assert lib.mlgnn_linear_bwd_supported(5000000, 128, 256, LN) == 1           # row slabs inside the entry point (>= 4 GiB operands)
    assert lib.mlgnn_linear_bwd_workspace_floats(1000, 128, 256, LN) == 256 * (128 * 256 + 128 + 512) + 768
    assert lib.mlgnn_linear_bwd_workspace_floats(1000, 256, 128, SHIFT) == 256 * (256 * 128 + 256) + 768
    assert lib.mlgnn_linear_bwd_workspace_floats(1000, 100, 256, LN) == -2
    call = lambda epi, M, K, ws=1 << 30: lib.mlgnn_linear_bwd(None, None, None, None, 0, None, epi, None, None, None, None,
                                                              None, None, None, None, None, None, ws, 1000, M, K, None)
    assert call(LN, 128, 256) == -1 and call(SHIFT, 256, 128) == -1             # NULL operands
    assert call(LN, 256, 128) == -2 and call(5, 128, 256) == -3
    # bf16 input gradient + rescaled cotangent
    assert lib.mlgnn_tallgemm_bf16_shift_supported(1000, 512, 256) == 1
    assert lib.mlgnn_tallgemm_bf16_shift_supported(1000, 40, 64) == 0 and lib.mlgnn_tallgemm_bf16_shift_supported(0, 512, 256) == 0
    sh = lambda N, R, J, ws=1 << 20: lib.mlgnn_tallgemm_bf16_shift(None, None, None, None, None, None, None, ws, N, R, J, None)
    assert sh(0, 512, 256) == 0 and sh(1000, 512, 256) == -1 and sh(1000, 40, 64) == -2


def test_argument_errors_of_the_fp32_large_diffpool_and_table_gradient():
    """mlgnn_diffpool_large_f32_* and mlgnn_table_grad_*: sizes, NULL operands and workspaces are checked before
    anything is launched (no GPU needed)."""
    from mlgnn import _lib
    lib = _lib.lib
    N, K, C = 256, 128, 128
    need = lib.mlgnn_diffpool_large_f32_workspace_bytes(N, K, C)
    saved = lib.mlgnn_diffpool_large_f32_saved_bytes(N, K, C)
    assert 0 < saved < need and need % 256 == 0
    assert lib.mlgnn_diffpool_large_f32_workspace_bytes(N, K + 64, C) == -2
    assert lib.mlgnn_diffpool_large_f32_bwd_workspace_bytes(N, K, C, 1) < lib.mlgnn_diffpool_large_f32_bwd_workspace_bytes(N, K, C, 0)
    fwd = lambda n=N, b=1, ws=need: lib.mlgnn_diffpool_large_f32_fwd(None, None, None, None, None, None, None, None, None,
                                                                      ws, n, K, C, b, 0, None)
    assert fwd() == -1 and fwd(n=100) == -2 and fwd(b=0) == -2 and fwd(b=70000) == -2
    bwd = lambda n=N: lib.mlgnn_diffpool_large_f32_bwd(None, None, None, None, None, None, None, None, None, None, None, 0,
                                                       None, 1 << 40, n, K, C, 1, 0, None)
    assert bwd() == -1 and bwd(n=130) == -2
    # fixed-point table gradient (max aggregator with a table edge term)
    assert lib.mlgnn_table_grad_bytes(20000, 128) == 256 + 20000 * 128 * 8
    assert lib.mlgnn_table_grad_bytes(10, 130) == -2 and lib.mlgnn_table_grad_bytes(0, 128) == -2
    assert lib.mlgnn_table_grad_begin(None, 100, 128, None, None) == -1
    assert lib.mlgnn_table_grad_begin(None, 0, 128, None, None) == -2 and lib.mlgnn_table_grad_begin(None, 10, 6, None, None) == -2
    assert lib.mlgnn_table_grad_finish(None, None, 10, 128, 0, None) == -1
    assert lib.mlgnn_table_grad_finish(None, None, 10, 127, 0, None) == -2


def test_supported_and_entry_points_agree_past_4_gib():
    """The dense backward entry points at row counts around and past the 4 GiB operand mark (4.19 M rows x 256 fp32
    columns; 5.12 M rows = 512 graphs of BASELINE configs[3] on one GPU): what ``*_supported`` / ``*_workspace_floats``
    promise is what the entry point accepts -- with NULL operands an accepted shape reports MLGNN_E_NULL (-1), a refused
    one MLGNN_E_SHAPE (-2); nothing is launched (no GPU needed)."""
    from mlgnn import _lib
    lib = _lib.lib
    LN, PLAIN, SHIFT = 0, 1, 2
    for N in (4_000_000, 4_194_304, 4_200_000, 5_120_000):
        for (M, K, epi) in ((128, 256, LN), (256, 128, SHIFT), (256, 128, PLAIN)):
            ok = lib.mlgnn_linear_bwd_supported(N, M, K, epi)
            ws = lib.mlgnn_linear_bwd_workspace_floats(N, M, K, epi)
            rc = lib.mlgnn_linear_bwd(None, None, None, None, 0, None, epi, None, None, None, None, None, None, None, None,
                                      None, None, 1 << 40, N, M, K, None)
            assert ok == 1 and ws > 0 and rc == -1, (N, M, K, epi, ok, ws, rc)
        for (M, K) in ((128, 256), (256, 128), (128, 128), (64, 128)):
            ws = lib.mlgnn_linear_wgrad_workspace_floats(N, M, K, 0)
            rc = lib.mlgnn_linear_wgrad(None, None, None, None, None, None, None, None, 1 << 40, N, M, K, 0, None)
            assert ws > 0 and rc == -1, (N, M, K, ws, rc)
        for (R, J) in ((128, 256), (256, 128), (128, 128)):
            ok = lib.mlgnn_tallgemm_lnbwd_supported(N, R, J)
            # (NULL grad_gamma_beta is the first pointer the entry point looks at, after every shape check)
            rc = lib.mlgnn_tallgemm_lnbwd(None, None, 1, None, None, None, None, None, None, None, None, None, 1 << 40,
                                          N, R, J, None)
            assert ok == 1 and rc == -1, (N, R, J, ok, rc)
            assert lib.mlgnn_tallgemm_supported(N, R, J, 0) == 1
            assert lib.mlgnn_tallgemm_nt_shift_supported(N, J, R) == lib.mlgnn_tallgemm_lnin_postln_supported(N, J, R)
    # refused shapes are refused by both
    assert lib.mlgnn_linear_bwd_supported(5_120_000, 256, 128, LN) == 0
    assert lib.mlgnn_linear_bwd(None, None, None, None, 0, None, LN, None, None, None, None, None, None, None, None, None,
                                None, 1 << 40, 5_120_000, 256, 128, None) == -2
    assert lib.mlgnn_tallgemm_lnbwd_supported(5_120_000, 512, 128) == 0
    assert lib.mlgnn_tallgemm_lnbwd(None, None, 1, None, None, None, None, None, None, None, None, None, 1 << 40,
                                    5_120_000, 512, 128, None) == -2
