"""Python mirror of the reference's solver interface (src/LyapunovSolverDecl.hpp:13-35) over the C ABI of
include/rails_solver.h: `Solver(A, B, M)`, `set_parameters(dict)`, `solve(V0=None)` -> (code, V, T).

Parameter names and defaults are the reference's (src/LyapunovSolver.hpp:27-36,76-87); return codes too
(0 converged, -1 not converged, 1 loop exhausted; 2 = stopped by the max_trips extension)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from ._solver_sigs import TRIP_FN
from .wrappers import HipOperatorWrapper, _f, _p


class Solver:
    def __init__(self, ctx, A, B, M=None, m_global=None):
        """A, M: HipOperatorWrapper (M None = identity).  B: host array, local rows x p."""
        self.ctx = ctx
        self.lib = ctx.lib
        self.A, self.M = A, M
        B = _f(B)
        self.m_local, self.p = B.shape
        h = C.c_void_p()
        check(self.lib.rails_solver_create(ctx.h, A.h.h, M.h.h if M is not None else None, _p(B), B.shape[0], B.shape[1],
                                           m_global if m_global is not None else -1, C.byref(h)), "rails_solver_create")
        self.h = h
        self._cb = None
        self.k = 0
        ctx._solvers.add(self)

    def set_parameters(self, params):
        for name, value in params.items():
            check(self.lib.rails_solver_set_parameter(self.h, name.encode(), float(value)), "rails_solver_set_parameter")
        code = C.c_int(0)
        check(self.lib.rails_solver_apply_parameters(self.h, C.byref(code)), "rails_solver_apply_parameters")
        return code.value

    def set_option(self, name, value):
        check(self.lib.rails_solver_set_option(self.h, name.encode(), float(value)), "rails_solver_set_option")

    def set_trip_callback(self, fn):
        if fn is None:
            self._cb = None
            check(self.lib.rails_solver_set_trip_callback(self.h, TRIP_FN(0), None), "rails_solver_set_trip_callback")
            return

        def tramp(user, trip):
            try:
                fn(trip)
            except Exception:
                import traceback
                traceback.print_exc()
        self._cb = TRIP_FN(tramp)
        check(self.lib.rails_solver_set_trip_callback(self.h, self._cb, None), "rails_solver_set_trip_callback")

    def solve(self, V0=None, fetch=True):
        if V0 is not None:
            V0 = _f(V0)
            check(self.lib.rails_solver_set_V(self.h, _p(V0), V0.shape[0], V0.shape[1]), "rails_solver_set_V")
        code, k = C.c_int(0), C.c_int(0)
        check(self.lib.rails_solver_solve(self.h, C.byref(code), C.byref(k)), "rails_solver_solve")
        self.k = k.value
        if not fetch:
            return code.value, None, None
        return code.value, self.V(), self.T()

    def V(self):
        V = np.zeros((self.m_local, self.k), order="F")
        check(self.lib.rails_solver_get_V(self.h, _p(V), max(1, self.m_local)), "rails_solver_get_V")
        return V

    def T(self):
        T = np.zeros((self.k, self.k), order="F")
        check(self.lib.rails_solver_get_T(self.h, _p(T), max(1, self.k)), "rails_solver_get_T")
        return T

    def trips(self):
        return self.lib.rails_solver_trips(self.h)

    def history(self):
        n = self.trips()
        out = np.zeros(max(n, 1))
        self.lib.rails_solver_history(self.h, _p(out), out.size)
        return out[:n]

    def profile(self):
        """Host wall-clock seconds per solver section of the last solve (reference's profile section names)."""
        import json
        buf = C.create_string_buffer(4096)
        check(self.lib.rails_solver_profile(self.h, buf, 4096), "rails_solver_profile")
        return json.loads(buf.value.decode())

    def relative_residual(self):
        """||A X M' + M X A' + B B'||_F / ||B B'||_F from Gram products on the device; bottoms out near 1e-8 (include/rails_solver.h)"""
        rel = C.c_double(0.0)
        check(self.lib.rails_solver_relative_residual(self.h, C.byref(rel)), "rails_solver_relative_residual")
        return rel.value

    def backend_stats(self):
        """counters of the coordinate-space back end for the last solve ({} when the direct back end ran)"""
        import json

        return json.loads(self.lib.rails_solver_backend_stats(self.h).decode())

    def close(self):
        if self.h:
            if self.ctx.h:  # after the context is gone the handle cannot be released safely any more
                self.lib.rails_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
