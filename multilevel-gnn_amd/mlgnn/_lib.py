"""ctypes binding of the C ABI in include/mlgnn.h.  Fails loudly when the library is missing."""
import ctypes
import os

# (MLGNN_LIB: a development override for same-box A/B runs of kernel variants, tools/build_variant.py)
LIB = os.environ.get("MLGNN_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmlgnn.so")

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_INT = _c.c_int
_F = _c.c_float

# name -> (restype, argtypes); mirrors include/mlgnn.h one to one (tests check the export list)
SIGNATURES = {
    "mlgnn_version": (_INT, []),
    "mlgnn_csr_aggregate_bwd_workspace_floats": (_I64, [_I64, _I64, _INT, _INT, _INT, _INT]),
    "mlgnn_csr_aggregate_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _I64, _I64, _INT, _INT, _INT, _INT, _INT, _F, _F, _P, _P, _F, _INT, _P, _P]),
    "mlgnn_csr_aggregate_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P, _P, _P, _I64,
                                       _I64, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _F, _F, _P, _P, _F, _INT, _INT, _P,
                                       _P, _P, _P]),
    "mlgnn_csr_aggregate_bwd_ln_workspace_floats": (_I64, [_I64, _I64]),
    "mlgnn_csr_aggregate_bwd_ln": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          _P, _P, _P, _P, _I64,
                                          _I64, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _F, _F, _P, _P, _F, _INT, _INT, _P,
                                          _P, _P, _P, _P]),
    "mlgnn_power_bwd_prologue_workspace_floats": (_I64, []),
    "mlgnn_power_bwd_prologue": (_INT, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_embedding_bwd": (_INT, [_P, _P, _P, _P, _I64, _I64, _INT, _P]),
    "mlgnn_max_table_grad_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_max_table_grad_workspace_floats": (_I64, [_I64, _I64, _I64]),
    "mlgnn_max_table_grad": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_max_table_grad_by_type": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_csr_aggregate_bwd_slots_offset_floats": (_I64, [_I64, _I64, _INT]),
    "mlgnn_max_sparse_supported": (_INT, [_I64, _I64]),
    "mlgnn_max_sparse_records": (_I64, [_I64, _I64, _I64]),
    "mlgnn_max_winners": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mlgnn_max_sparse_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mlgnn_max_sparse_table_grad": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_table_grad_bytes": (_I64, [_I64, _I64]),
    "mlgnn_table_grad_begin": (_INT, [_P, _I64, _I64, _P, _P]),
    "mlgnn_table_grad_finish": (_INT, [_P, _P, _I64, _I64, _INT, _P]),
    "mlgnn_segment_project_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_segment_project_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                         _I64, _I64, _I64, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_layernorm_bwd_workspace_floats": (_I64, [_I64, _I64, _INT]),
    "mlgnn_layernorm_act_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I64, _I64, _F, _INT, _INT, _P]),
    "mlgnn_layernorm_act_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _F, _I64, _I64, _INT, _INT, _P]),
    "mlgnn_linear_wgrad_workspace_floats": (_I64, [_I64, _I64, _I64, _INT]),
    "mlgnn_tallgemm_bf16_shift_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_tallgemm_bf16_shift": (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_linear_bwd_supported": (_INT, [_I64, _I64, _I64, _INT]),
    "mlgnn_linear_bwd_workspace_floats": (_I64, [_I64, _I64, _I64, _INT]),
    "mlgnn_linear_bwd": (_INT, [_P, _P, _P, _P, _INT, _P, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64,
                                _I64, _I64, _I64, _P]),
    "mlgnn_linear_wgrad": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_diffpool_fwd_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_diffpool_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _P]),
    "mlgnn_coo_to_csr_workspace_bytes": (_I64, [_I64, _I64]),
    "mlgnn_coo_to_csr": (_INT, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mlgnn_msgnorm_bwd_workspace_floats": (_I64, [_I64, _I64]),
    "mlgnn_msgnorm_add_fwd": (_INT, [_P, _P, _P, _P, _I64, _I64, _INT, _P]),
    "mlgnn_msgnorm_add_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_diffpool_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _P]),
    "mlgnn_edge_table_to_csr": (_INT, [_P, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _P]),
    "mlgnn_dense_sage_supported": (_INT, [_I64, _I64, _I64, _INT]),
    "mlgnn_dense_sage_bwd_workspace_floats": (_I64, [_I64, _I64, _I64]),
    "mlgnn_dense_sage_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P]),
    "mlgnn_dense_sage_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64,
                                    _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P]),
    "mlgnn_segment_pool_workspace_bytes": (_I64, [_I64, _I64]),
    "mlgnn_segment_pool_fwd": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _INT, _P]),
    "mlgnn_tallgemm_supported": (_INT, [_I64, _I64, _I64, _INT]),
    "mlgnn_tallgemm_workspace_bytes": (_I64, [_I64, _I64, _INT]),
    "mlgnn_tallgemm_nt": (_INT, [_P, _P, _INT, _P, _P, _P, _INT, _P, _P, _F, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_tallgemm_nt_shift_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_tallgemm_nt_shift": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_tallgemm_lnin_postln_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_tallgemm_lnin_postln": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _INT, _P, _P, _P, _P, _P, _I64,
                                          _I64, _I64, _I64, _P]),
    "mlgnn_tallgemm_lnbwd_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_tallgemm_lnbwd_workspace_bytes": (_I64, [_I64, _I64]),
    "mlgnn_tallgemm_lnbwd": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_csr_replicate": (_INT, [_P] * 14 + [_I64, _I64, _I64, _P]),
    "mlgnn_sage_rewrite": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "mlgnn_tallgemm_dual_supported": (_INT, [_I64, _I64, _I64, _I64]),
    "mlgnn_tallgemm_dual": (_INT, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_transpose_batched": (_INT, [_P, _P, _I64, _I64, _I64, _P]),
    "mlgnn_sage_fold_fwd": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _INT, _P]),
    "mlgnn_sage_fold_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _INT, _P]),
    "mlgnn_leaky_relu_bwd": (_INT, [_P, _P, _P, _F, _P, _P, _I64, _I64, _P]),
    "mlgnn_node_embed_fwd": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "mlgnn_node_embed_bwd": (_INT, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "mlgnn_narrow_linear_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_narrow_linear_bwd_workspace_floats": (_I64, [_I64, _I64]),
    "mlgnn_narrow_linear_fwd": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "mlgnn_narrow_linear_bwd": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_skinny_linear_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_skinny_linear_fwd_workspace_floats": (_I64, [_I64, _I64, _I64]),
    "mlgnn_skinny_linear_fwd": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_skinny_linear_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "mlgnn_stream_copy": (_INT, [_P, _P, _I64, _INT, _P]),
    "mlgnn_gemm_bf16_nt_workgroups": (_INT, [_I64, _I64, _INT]),
    "mlgnn_gemm_bf16_nt": (_INT, [_c.POINTER(_P), _c.POINTER(_P), _c.POINTER(_I64), _c.POINTER(_I64), _c.POINTER(_I64),
                                  _INT, _I64, _I64, _INT, _P, _P, _I64, _INT, _P, _I64, _P, _I64, _INT, _F,
                                  _P, _I64, _P, _P]),
    "mlgnn_diffpool_large_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_diffpool_large_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_diffpool_large_saved_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_diffpool_large_fwd": (_INT, [_P, _P, _P, _INT, _P, _P, _P, _P, _INT, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_diffpool_large_bwd_workspace_bytes": (_I64, [_I64, _I64, _I64, _INT]),
    "mlgnn_diffpool_large_bwd": (_INT, [_P, _P, _P, _INT, _P, _P, _P, _P, _INT, _P, _P, _INT, _P, _P, _P, _P, _INT, _P, _I64,
                                        _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_diffpool_large_f32_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_diffpool_large_f32_saved_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_diffpool_large_f32_fwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _P]),
    "mlgnn_diffpool_large_f32_bwd_workspace_bytes": (_I64, [_I64, _I64, _I64, _INT]),
    "mlgnn_diffpool_large_f32_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _INT, _P, _I64, _I64, _I64, _I64,
                                            _I64, _INT, _P]),
    "mlgnn_linear_f32x3_supported": (_INT, [_I64, _I64, _I64]),
    "mlgnn_linear_f32x3_padded_rows": (_I64, [_I64]),
    "mlgnn_linear_f32x3_fwd_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_linear_f32x3_fwd": (_INT, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_linear_f32x3_bwd_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "mlgnn_linear_f32x3_bwd": (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mlgnn_adam_workspace_floats": (_I64, []),
    "mlgnn_adam_step": (_INT, [_P, _P, _P, _P, _I64, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "mlgnn_hub_capacity": (_I64, [_I64, _INT]),
    "mlgnn_hub_scratch_bytes": (_I64, [_I64, _I64]),
    "mlgnn_hub_rows": (_INT, [_P, _I64, _INT, _I64, _P, _P, _P, _P]),
    "mlgnn_canary_malloc": (_P, [_I64, _INT, _P]),
    "mlgnn_canary_free": (None, [_P, _I64, _INT, _P]),
    "mlgnn_canary_check": (_I64, [_c.c_char_p, _I64]),
    "mlgnn_canary_stats": (_I64, [_c.POINTER(_I64)]),
}


class HubStruct(_c.Structure):
    """``mlgnn_hub_t`` (include/mlgnn.h)."""
    _fields_ = [("cap", _c.c_int32), ("capacity", _c.c_int32), ("vrows", _P), ("hubs", _P), ("counts", _P),
                ("tmp", _P), ("tmp_bytes", _I64)]


class LnFoldStruct(_c.Structure):
    """``mlgnn_ln_fold_t`` (include/mlgnn.h)."""
    _fields_ = [("h", _P), ("mean", _P), ("rstd", _P), ("gamma", _P), ("beta", _P), ("grad_extra", _P), ("row_max", _P),
                ("grad_gamma_beta", _P), ("workspace", _P), ("workspace_floats", _I64), ("relu", _c.c_int32)]


ERRORS = {-1: "MLGNN_E_NULL", -2: "MLGNN_E_SHAPE", -3: "MLGNN_E_MODE", -4: "MLGNN_E_DTYPE",
          -5: "MLGNN_E_WORKSPACE", -6: "MLGNN_E_ALIGN"}


class MlgnnError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB):
        raise ImportError(
            "libmlgnn.so is not built (%s). Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB)
    lib = ctypes.CDLL(LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the export is missing
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()

# MLGNN_CANARY=1 (debug; tests/README): every device allocation of the process gets guard bands (csrc/canary.hip
# becomes torch's allocator -- this must happen before the first device allocation) and every C-ABI call is followed by
# a device synchronise + a comparison of all bands; an out-of-bounds write is reported at the call that made it.
CANARY = os.environ.get("MLGNN_CANARY", "0") == "1"
CANARY_CALLS = 0


def canary_check(what):
    """Synchronise and compare every guard band; raises :class:`MlgnnError` naming ``what`` on damage."""
    buf = ctypes.create_string_buffer(512)
    rc = lib.mlgnn_canary_check(buf, 512)
    if rc != 0:
        raise MlgnnError("MLGNN_CANARY: after %s: %s" % (what, buf.value.decode(errors="replace") or "check failed (%d)" % rc))


def canary_stats():
    out = (_I64 * 4)()
    lib.mlgnn_canary_stats(out)
    return dict(live=out[0], device_allocations=out[1], reuses=out[2], checks=out[3], guarded_calls=CANARY_CALLS)


class _CanaryLib:
    """``lib`` with a guard-band check behind every call that takes a stream (the ones that launch kernels)."""

    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, name):
        fn = getattr(self._inner, name)
        if name.startswith("mlgnn_canary") or name not in SIGNATURES or not SIGNATURES[name][1] \
                or SIGNATURES[name][0] is not _INT or name.endswith("_supported") or name.endswith("_workgroups"):
            return fn

        def guarded(*args):
            global CANARY_CALLS
            rc = fn(*args)
            CANARY_CALLS += 1
            canary_check(name)
            return rc
        setattr(self, name, guarded)
        return guarded


def _install_canary():
    import torch
    alloc = torch.cuda.memory.CUDAPluggableAllocator(LIB, "mlgnn_canary_malloc", "mlgnn_canary_free")
    torch.cuda.memory.change_current_allocator(alloc)       # raises if the process has allocated device memory already
    return alloc


if CANARY:
    _CANARY_ALLOCATOR = _install_canary()
    lib = _CanaryLib(lib)


def check(code, what):
    if code == 0:
        return
    if code < 0:
        raise MlgnnError("%s: argument error %s" % (what, ERRORS.get(code, code)))
    raise MlgnnError("%s: HIP error %d" % (what, code))


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
