"""ctypes binding of librails_hip.so (include/rails_hip.h, include/rails_solver.h).

The library is built in-tree (rails_amd/lib/librails_hip.so) by rails_amd.build.build() /
`make -C rails_amd/csrc`.  There is no CPU fallback: if the shared library is missing, or no
gfx950 device is visible when a context is created, the calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librails_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int)

# name -> (restype, argtypes); every symbol include/rails_hip.h declares
SIGNATURES = {
    "rails_last_error": (C.c_char_p, []),
    "rails_version": (C.c_char_p, []),
    "rails_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "rails_ctx_destroy": (C.c_int, [_vp]),
    "rails_ctx_sync": (C.c_int, [_vp]),
    "rails_ctx_stream": (_vp, [_vp]),
    "rails_ctx_set_meter": (C.c_int, [_vp, C.c_int]),
    "rails_ctx_enable_library_gemm": (C.c_int, [_vp]),
    "rails_ctx_library_gemm_ready": (C.c_int, [_vp]),
    "rails_deferred_reserve": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "rails_gram_deferred": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "rails_update_gram_deferred": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "rails_panel_gemm_deferred": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int]),
    "rails_chol_inverse_deferred": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "rails_deferred_fetch": (C.c_int, [_vp, C.c_int, C.c_int64, _dp]),
    "rails_ctx_stats": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "rails_ctx_set_seed": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "rails_ctx_rng_state": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rails_ctx_set_partition": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "rails_ctx_set_allreduce": (C.c_int, [_vp, ALLREDUCE_FN, _vp]),
    "rails_rccl_unique_id": (C.c_int, [_vp]),
    "rails_ctx_init_rccl": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rails_ctx_set_rccl": (C.c_int, [_vp, _vp]),
    "rails_ctx_rccl_size": (C.c_int, [_vp]),
    "rails_csr_set_halo_counts": (C.c_int, [_vp, C.c_int, _i64p, _i64p]),
    "rails_csr_create": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, _i32p, _dp, C.POINTER(_vp)]),
    "rails_csr_create_rect": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, _i32p, _dp, C.POINTER(_vp)]),
    "rails_csr_create_callback": (C.c_int, [_vp, C.c_int64, APPLY_FN, _vp, C.POINTER(_vp)]),
    "rails_csr_destroy": (C.c_int, [_vp]),
    "rails_csr_rows": (C.c_int64, [_vp]),
    "rails_csr_nnz": (C.c_int64, [_vp]),
    "rails_csr_set_halo": (C.c_int, [_vp, C.c_int64, _i64p, C.c_int64, HALO_FN, _vp]),
    "rails_spmm": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int]),
    "rails_csr_set_variant": (C.c_int, [_vp, C.c_int]),
    "rails_csr_sweep_stats": (C.c_int, [_vp, C.c_int, _dp]),
    "rails_csr_prepare": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rails_csr_last_kernel": (C.c_char_p, [_vp]),
    "rails_sweep_plan_create": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i32p, _dp, _ip, C.POINTER(_vp)]),
    "rails_sweep_plan_destroy": (C.c_int, [_vp]),
    "rails_sweep_plan_info": (C.c_int, [_vp, _i64p, _dp]),
    "rails_sweep_plan_array": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _i64p]),
    "rails_panel_create": (C.c_int, [_vp, C.c_int64, C.c_int, C.POINTER(_vp)]),
    "rails_panel_destroy": (C.c_int, [_vp]),
    "rails_panel_rows": (C.c_int64, [_vp]),
    "rails_panel_capacity": (C.c_int, [_vp]),
    "rails_panel_ld": (C.c_int, [_vp]),
    "rails_panel_device_ptr": (_vp, [_vp]),
    "rails_panel_reserve": (C.c_int, [_vp, _vp, C.c_int]),
    "rails_panel_upload": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp, C.c_int64]),
    "rails_panel_download": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp, C.c_int64]),
    "rails_panel_fill": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double]),
    "rails_panel_scale": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double]),
    "rails_panel_copy": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int]),
    "rails_panel_axpy": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, _vp, C.c_int]),
    "rails_sptrsv_create": (C.c_int, [_vp, C.c_int64, _i64p, _i32p, _dp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rails_sptrsv_destroy": (None, [_vp]),
    "rails_sptrsv_levels": (C.c_int64, [_vp]),
    "rails_sptrsv_solve": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "rails_panel_permute_rows": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "rails_index_upload": (C.c_int, [_vp, _i32p, C.c_int64, C.POINTER(_vp)]),
    "rails_index_free": (None, [_vp, _vp]),
    "rails_panel_random": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rails_gram": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _dp, C.c_int]),
    "rails_panel_gemm": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_double, _vp, C.c_int]),
    "rails_panel_gemm_wide": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_double, _vp, C.c_int]),
    "rails_orthogonalize": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _ip]),
    "rails_resid_lanczos": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                      _dp, C.c_int, _ip]),
    "rails_lanczos_start": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _dp]),
    "rails_lanczos_vectors": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _vp, C.c_int]),
    "rails_lanczos_release": (C.c_int, [_vp]),
    "rails_timer_start": (C.c_int, [_vp]),
    "rails_timer_stop": (C.c_int, [_vp, _dp]),
    "rails_sb03md": (None, [C.c_char, C.c_char, C.c_char, C.c_char, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _ip]),
    "rails_dsyev": (None, [C.c_char, C.c_char, C.c_int, _dp, C.c_int, _dp, _ip]),
    "rails_dsteqr": (None, [C.c_char, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _ip]),
    "rails_ctx_reserve_staging": (C.c_int, [_vp, C.c_size_t]),
    "rails_sb03md_set_pause": (None, [C.c_int]),
    "rails_sb03md_counts": (None, [C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "rails_sb03md_adi_counts": (None, [C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "rails_dtrsm": (None, [C.c_char, C.c_char, C.c_char, C.c_char, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int]),
    "rails_dgemm": (None, [C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int]),
    "rails_dpotrf": (None, [C.c_char, C.c_int, _dp, C.c_int, _ip]),
    "rails_dpstrf": (None, [C.c_char, C.c_int, _dp, C.c_int, _ip, _ip, C.c_double, _ip]),
    "rails_range_basis": (None, [C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int, _ip, _ip]),
    "rails_host_lapack_init": (C.c_int, [C.c_char_p]),
    "rails_host_lapack_path": (C.c_char_p, []),
}

_lib = None


class RailsError(RuntimeError):
    pass


def load():
    """Load librails_hip.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RailsError(
            "rails_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rails_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    try:
        from ._solver_sigs import bind as _bind_solver
        _bind_solver(lib)
    except ImportError:
        pass
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().rails_last_error()
        raise RailsError("%s failed (code %d): %s" % (what or "rails call", rc, msg.decode() if msg else ""))
