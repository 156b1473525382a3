"""ctypes signatures of include/rails_solver.h."""
import ctypes as C

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

TRIP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int)

SOLVER_SIGNATURES = {
    "rails_solver_create": (C.c_int, [_vp, _vp, _vp, _dp, C.c_int64, C.c_int, C.c_int64, C.POINTER(_vp)]),
    "rails_solver_destroy": (C.c_int, [_vp]),
    "rails_solver_set_parameter": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "rails_solver_apply_parameters": (C.c_int, [_vp, _ip]),
    "rails_solver_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "rails_solver_set_trip_callback": (C.c_int, [_vp, TRIP_FN, _vp]),
    "rails_solver_set_V": (C.c_int, [_vp, _dp, C.c_int64, C.c_int]),
    "rails_solver_solve": (C.c_int, [_vp, _ip, _ip]),
    "rails_solver_get_V": (C.c_int, [_vp, _dp, C.c_int64]),
    "rails_solver_get_T": (C.c_int, [_vp, _dp, C.c_int]),
    "rails_solver_trips": (C.c_int, [_vp]),
    "rails_solver_history": (C.c_int, [_vp, _dp, C.c_int]),
    "rails_solver_relative_residual": (C.c_int, [_vp, _dp]),
    "rails_solver_profile": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "rails_solver_backend_stats": (C.c_char_p, [_vp]),
}


def bind(lib):
    for name, (res, args) in SOLVER_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
