"""ctypes binding of libgcnspmm.so (the C ABI declared in include/gcn_spmm.h).

The product path has NO CPU fallback: if the library is missing or a call returns a
non-zero status, a GcnAmdError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GCN_AMD_LIB: load a private copy — long-running host tools that must survive a rebuild of the in-tree library)
LIB_PATH = os.environ.get("GCN_AMD_LIB") or os.path.join(_HERE, "lib", "libgcnspmm.so")
DROPIN_DIR = os.path.join(_HERE, "dropin")


class GcnAmdError(RuntimeError):
    status = None            # the C-ABI status code when the error came from a call (include/gcn_spmm.h), else None


ERR_NOT_FACTORED = 6         # GCN_ERR_NOT_FACTORED
ERR_INTERNAL = 7             # GCN_ERR_INTERNAL (a consistency guard tripped: gcn_order_rabbit_device)


_c_i32 = ctypes.c_int32
_c_p = ctypes.c_void_p

# name -> (restype, argtypes)  — one entry per symbol declared in include/gcn_spmm.h
SIGNATURES = {
    "gcn_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "gcn_version": (ctypes.c_char_p, []),
    "gcn_device_cu_count": (ctypes.c_int, []),
    "gcn_spmm_plan_create": (ctypes.c_int, [ctypes.POINTER(_c_p), _c_p, _c_i32, _c_i32, _c_i32, _c_i32, _c_p]),
    "gcn_spmm_plan_destroy": (ctypes.c_int, [_c_p]),
    "gcn_spmm_plan_num_chunks": (_c_i32, [_c_p]),
    "gcn_spmm_plan_chunk_nnz": (_c_i32, [_c_p]),
    "gcn_spmm_plan_workspace_bytes": (ctypes.c_size_t, [_c_p, _c_i32]),
    "gcn_spmm_csr_f32": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i32, _c_p]),
    "gcn_spmm_csr_f32_bias_relu": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i32, _c_i32, _c_p]),
    "gcn_spmm_csr_f32_epilogue": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i32, ctypes.c_float,
                                                 ctypes.c_uint64, ctypes.c_uint64, _c_i32, _c_p]),
    "gcn_dropout_f32": (ctypes.c_int, [_c_p, _c_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_p]),
    "gcn_spmm_plan_set_tile_cols": (ctypes.c_int, [_c_p, _c_i32]),
    "gcn_spmm_plan_set_blocks_per_cu": (ctypes.c_int, [_c_p, _c_i32]),
    "gcn_spmm_plan_set_gather_width": (ctypes.c_int, [_c_p, _c_i32]),
    "gcn_exchange_flags_create": (ctypes.c_int, [_c_i32, _c_p, _c_p]),
    "gcn_exchange_flags_open": (ctypes.c_int, [_c_p, _c_p]),
    "gcn_exchange_flags_close": (ctypes.c_int, [_c_p]),
    "gcn_exchange_flags_destroy": (ctypes.c_int, [_c_p]),
    "gcn_exchange_push": (ctypes.c_int, [_c_p, _c_p, ctypes.c_size_t, _c_p]),
    "gcn_exchange_signal": (ctypes.c_int, [_c_p, _c_p, _c_p]),
    "gcn_exchange_wait": (ctypes.c_int, [_c_p, _c_i32, _c_i32, _c_i32, _c_p, ctypes.c_double, _c_p]),
    "gcn_spmm_plan_num_passes": (_c_i32, [_c_p, _c_i32]),
    "gcn_spmm_plan_main_kernel": (ctypes.c_int, [_c_p, _c_i32, _c_i32, ctypes.c_char_p, _c_i32]),
    "gcn_spmm_plan_enable_slicing": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_i32, _c_p]),
    "gcn_spmm_plan_num_slices": (_c_i32, [_c_p]),
    "gcn_spmm_plan_narrow_slices": (_c_i32, [_c_p, _c_i32]),
    "gcn_spmm_plan_prepare_width": (_c_i32, [_c_p, _c_p, _c_p, _c_p, _c_i32, _c_p]),
    "gcn_spmm_plan_set_value_factors": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "gcn_spmm_plan_has_value_factors": (_c_i32, [_c_p]),
    "gcn_spmm_plan_enable_panels": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_i32, _c_p]),
    "gcn_spmm_plan_panel_rows": (_c_i32, [_c_p]),
    "gcn_spmm_plan_panel_coverage": (ctypes.c_double, [_c_p]),
    "gcn_spmm_plan_dense_panels": (_c_i32, [_c_p]),
    "gcn_spmm_profile_begin": (ctypes.c_int, [_c_p, _c_i32]),
    "gcn_spmm_profile_end": (ctypes.c_int, [_c_p, _c_p, _c_p]),
    "gcn_spmm_csr_f32_oneshot": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i32, _c_i32, _c_i32, _c_i32, _c_p]),
    "gcn_spmm_plan_prelaid_layout": (ctypes.c_int, [_c_p, _c_i32, _c_p, _c_p, _c_p, _c_p]),
    "gcn_spmm_csr_f32_prelaid": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i32, _c_i32, _c_p]),
    "gcn_spmm_auto_slices": (_c_i32, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _c_i32]),
    "gcn_spmm_group_addressing": (_c_i32, [ctypes.c_int64, _c_i32]),
    "gcn_gather_rows_f32": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_i32, _c_i32, _c_p]),
    "gcn_order_deg_device": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_i32, _c_i32, _c_p, _c_p]),
    "gcn_order_rcm_device": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_p, _c_p, _c_p]),
    "gcn_csr_apply_rank_device": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_p, _c_i32, _c_i32, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "gcn_order_rabbit_device": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_p, _c_p, _c_p, _c_p]),
    "gcn_order_rabbit": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_p, _c_p]),
    "gcn_order_deg": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_i32, _c_i32, _c_p]),
    "gcn_order_rcm": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_i32, _c_p]),
    "gcn_order_gorder": (ctypes.c_int, [_c_p, _c_p, _c_i32, _c_i32, _c_i32, _c_p]),
    "gcn_csr_apply_rank": (ctypes.c_int, [_c_p, _c_p, _c_p, _c_i32, _c_i32, _c_p, _c_p]),
    # drop-in symbols (reference signatures; void)
    "dfs": (None, [_c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "gorder": (None, [_c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "perm_apply": (None, [_c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "rabbit": (None, [_c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "csr2seg_Cmajor": (None, [ctypes.c_int, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_p, _c_p, _c_p, _c_p,
                              ctypes.c_int, _c_p]),
    "csr2tile": (None, [_c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_p, _c_p, _c_p,
                        _c_p, _c_p, _c_p, ctypes.c_int, _c_p]),
    "flexspmm": (None, [_c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                        ctypes.c_int, _c_p, _c_p]),
    "permutate": (None, [_c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "cuspmm": (None, [_c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
}

_lib = None


def load(path=None):
    """Load (once) and return the ctypes handle; raises GcnAmdError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise GcnAmdError(
            f"{p} not found: the HIP library is not built. Run `python -m gcn_amd.build` "
            "(there is no CPU fallback for the SpMM path).")
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # pragma: no cover - environment dependent
        raise GcnAmdError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GcnAmdError(f"{p} does not export `{name}` declared in include/gcn_spmm.h") from e
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().gcn_status_string(status).decode()
        err = GcnAmdError(f"{what}: {msg} (status {status})")
        err.status = int(status)
        raise err
