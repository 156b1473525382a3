"""Host-side schedule of the sweep SpMM kernel (rails_amd/csrc/sweep_plan.h) as numpy arrays: diagnostics and tests.

No device is touched: the plan is a function of the sparsity pattern and the geometry alone.
"""
import ctypes as C

import numpy as np

from . import _lib

_ARRAYS = [("part_row0", np.int64), ("sweep0", np.int64), ("nsteps", np.int32), ("hdr_off", np.int64), ("batch_off", np.int64),
           ("flush_off", np.int64), ("codes", np.uint32), ("vals", np.float64), ("offs", np.uint16), ("flush_rows", np.int32)]


class SweepPlan:
    """params = (waves, groups, rows per step, ring segments, parts, phases[, segments being filled[, trips per entry (4 or 2)]]) or None for
    the kernel's geometry."""

    def __init__(self, rowptr, col, val, ncols=None, params=None):
        lib = _lib.load()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        m = rowptr.size - 1
        self.m = m
        self.ncols = int(m if ncols is None else ncols)
        prm = None
        if params is not None:
            params = list(params)
            params += [1, 0][len(params) - 6:] if len(params) < 8 else []  # defaults: one segment being filled, the library's trips per entry
            prm = (C.c_int * 8)(*[int(v) for v in params])
        h = C.c_void_p()
        _lib.check(lib.rails_sweep_plan_create(m, self.ncols, rowptr.ctypes.data_as(_lib._i64p), col.ctypes.data_as(_lib._i32p),
                                               val.ctypes.data_as(_lib._dp), prm, C.byref(h)), "rails_sweep_plan_create")
        self._h = h
        ii = (C.c_int64 * 16)()
        dd = (C.c_double * 4)()
        _lib.check(lib.rails_sweep_plan_info(h, ii, dd), "rails_sweep_plan_info")
        self.iinfo = np.array(list(ii), dtype=np.int64)
        self.waves, self.groups, self.seg_rows, self.nseg, self.parts, self.phases, self.codes_per_step, self.trips, self.nnz, self.batches = [int(v) for v in ii[:10]]
        self.slots = int(ii[11])
        self.ahead = int(ii[12])
        self.entry_trips = int(ii[13])
        self.efficiency, self.staged_rows_per_row = dd[0], dd[1]
        for which, (name, dt) in enumerate(_ARRAYS):
            p = C.c_void_p()
            n = C.c_int64()
            _lib.check(lib.rails_sweep_plan_array(h, which, C.byref(p), C.byref(n)), "rails_sweep_plan_array")
            if n.value:
                buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p.value)
                arr = np.frombuffer(buf, dtype=dt)  # borrowed: valid until close()
            else:
                arr = np.zeros(0, dtype=dt)
            setattr(self, name, arr)

    def close(self):
        if self._h:
            for name, _ in _ARRAYS:
                setattr(self, name, None)
            _lib.load().rails_sweep_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
