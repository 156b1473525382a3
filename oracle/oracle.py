"""ctypes loader for the CPU oracle (oracle/_build/librails_oracle.so) and, when it has been
built in the build container, the compiled reference (oracle/_ref/librails_ref.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product package (rails_amd) never does.

Arrays cross the boundary column-major (Fortran order), float64, like the reference's
StlWrapper storage (src/StlVector.cpp:47-50).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "librails_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "librails_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


def build(ref=True):
    """Compile the oracle (and, if /root/reference is present, the reference build)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _f(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


def _p(a):
    return a.ctypes.data_as(_dp)


class Params(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("expand_size", C.c_int),
        ("lanczos_iterations", C.c_int),
        ("restart_size", C.c_int),
        ("reduced_size", C.c_int),
        ("restart_iterations", C.c_int),
        ("restart_tolerance", C.c_double),
        ("minimize_solution_space", C.c_int),
        ("restart_from_solution", C.c_int),
        ("rng_mode", C.c_int),
        ("seed", C.c_ulonglong),
        ("stream0", C.c_ulonglong),
        ("row0", C.c_long),
        ("verbose", C.c_int),
        ("max_trips", C.c_int),
    ]


_PARAM_KEYS = {
    "Maximum iterations": "max_iter",
    "Tolerance": "tol",
    "Expand size": "expand_size",
    "Lanczos iterations": "lanczos_iterations",
    "Restart size": "restart_size",
    "Reduced size": "reduced_size",
    "Restart iterations": "restart_iterations",
    "Restart tolerance": "restart_tolerance",
    "Minimize solution space": "minimize_solution_space",
    "Restart from solution": "restart_from_solution",
}


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.orc_lapack_init.argtypes = [C.c_char_p]
        L.orc_lapack_init.restype = C.c_int
        L.orc_lapack_path.restype = C.c_char_p
        L.orc_norm2.restype = C.c_double
        L.orc_norm2.argtypes = [C.c_int, C.c_int, _dp, C.c_int]
        L.orc_sb03md.argtypes = [C.c_char, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp]
        L.orc_dense_solve.argtypes = [C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        L.orc_dsyev.argtypes = [C.c_int, _dp, C.c_int, _dp]
        L.orc_dot.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        L.orc_panel_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int,
                                     C.c_double, _dp, C.c_int]
        L.orc_orthogonalize.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int]
        L.orc_csr_spmm.argtypes = [C.c_int, _i64p, _i32p, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        L.orc_find_largest.argtypes = [_dp, C.c_int, C.c_int, _ip]
        L.orc_random.argtypes = [C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_long, C.c_int, C.c_int, _dp, C.c_int]
        L.orc_resid_lanczos.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_long, _dp, _dp,
                                        _dp, _dp]
        L.orc_compute_restart_vectors.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_double, _dp]
        L.orc_solve.argtypes = [C.c_int, _dp, C.c_int, _i64p, _i32p, _dp, _i64p, _i32p, _dp, _dp, C.c_int, C.c_int,
                                C.POINTER(Params), _dp, C.c_int, C.c_int, _ip, _dp, C.c_int, _dp, C.c_int, _ip]
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_srand.argtypes = [C.c_uint]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        if L.orc_lapack_init(os.environ.get("RAILS_LAPACK_LIB", "").encode()) != 0:
            raise RuntimeError("oracle: no LAPACK library found")
        # OpenMP threads: never more than the CPUs this process may use (a GPU box gives one GPU's share of the host,
        # 16 CPUs; 128 spinning threads on 16 cores turn every parallel region into milliseconds)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cap = int(os.environ.get("RAILS_ORACLE_THREADS", "16"))
        L.orc_set_num_threads(max(1, min(avail, cap)))

    # ---- small helpers -------------------------------------------------------------------
    def params(self, d=None, **kw):
        p = Params()
        self.lib.orc_default_params(C.byref(p))
        tol_given = False
        rt_given = False
        for k, v in {**(d or {}), **kw}.items():
            key = _PARAM_KEYS.get(k, k)
            if key == "tol":
                tol_given = True
            if key == "restart_tolerance":
                rt_given = True
            setattr(p, key, type(getattr(p, key))(v))
        if tol_given and not rt_given:
            p.restart_tolerance = p.tol * 1e-3  # src/LyapunovSolver.hpp:83
        return p

    def srand(self, s):
        self.lib.orc_srand(s)

    def set_partition(self, allreduce=None, halo=None, plan=None, m_global=0):
        """Row-partitioned mode for the world_size > 1 CPU tests: allreduce(ptr, n, stream) and
        halo(send_ptr, recv_ptr, ncols, stream) are the hooks of rails_amd.partition (host buffers here)."""
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)
        HL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        self._ar = AR((lambda buf, n: int(allreduce(buf or 0, n, 0) or 0))) if allreduce else C.cast(None, AR)
        self._hl = HL((lambda s, r, nc: int(halo(s or 0, r or 0, nc, 0) or 0))) if halo else C.cast(None, HL)
        rows = np.ascontiguousarray(plan.send_rows if plan is not None else np.zeros(0), dtype=np.int64)
        self.lib.orc_set_partition.argtypes = [AR, HL, _i64p, C.c_int64, C.c_int64, C.c_int64]
        self.lib.orc_set_partition(self._ar, self._hl, rows.ctypes.data_as(_i64p), rows.size, plan.n_ghost if plan is not None else 0,
                                   m_global)

    def num_threads(self):
        return self.lib.orc_num_threads()

    def set_num_threads(self, n):
        self.lib.orc_set_num_threads(n)

    def random(self, m, n, mode=1, seed=1, stream=0, row0=0):
        X = np.zeros((m, n), order="F")
        self.lib.orc_random(mode, seed, stream, row0, m, n, _p(X), m)
        return X

    def dot(self, X, Y):
        X, Y = _f(X), _f(Y)
        Cm = np.zeros((X.shape[1], Y.shape[1]), order="F")
        self.lib.orc_dot(X.shape[0], X.shape[1], Y.shape[1], _p(X), X.shape[0], _p(Y), Y.shape[0], _p(Cm),
                         max(1, Cm.shape[0]))
        return Cm

    def panel_gemm(self, X, Cm, Y=None, alpha=1.0, beta=0.0):
        X, Cm = _f(X), _f(Cm)
        Yo = np.zeros((X.shape[0], Cm.shape[1]), order="F") if Y is None else _f(Y).copy(order="F")
        self.lib.orc_panel_gemm(X.shape[0], X.shape[1], Cm.shape[1], alpha, _p(X), X.shape[0], _p(Cm),
                                max(1, Cm.shape[0]), beta, _p(Yo), Yo.shape[0])
        return Yo

    def norm2(self, X):
        X = _f(X)
        return self.lib.orc_norm2(X.shape[0], X.shape[1], _p(X), X.shape[0])

    def orthogonalize(self, V, start=0):
        V = _f(V).copy(order="F")
        self.lib.orc_orthogonalize(V.shape[0], _p(V), V.shape[0], start, V.shape[1])
        return V

    def csr_spmm(self, rowptr, col, val, X):
        X = _f(X)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        m = rowptr.size - 1
        Y = np.zeros((m, X.shape[1]), order="F")
        self.lib.orc_csr_spmm(m, rowptr.ctypes.data_as(_i64p), col.ctypes.data_as(_i32p), _p(val), X.shape[1], _p(X),
                              X.shape[0], _p(Y), m)
        return Y

    def sweep_interpret(self, plan, X, n_chunks):
        """Run the sweep-SpMM schedule `plan` (rails_amd.sweep.SweepPlan) on the CPU: returns (rc, Y)."""
        X = _f(X)
        Y = np.full((plan.m, 16 * n_chunks), np.nan, order="F")
        u16p = C.POINTER(C.c_uint16)
        u32p = C.POINTER(C.c_uint32)
        fn = self.lib.orc_sweep_interpret
        fn.restype = C.c_int
        fn.argtypes = [_i64p, _i64p, _i64p, _i32p, _i64p, _i64p, _i64p, u32p, _dp, u16p, _i32p, C.c_int64, C.c_int64, C.c_int, _dp, C.c_int64,
                       _dp, C.c_int64]
        rc = fn(plan.iinfo.ctypes.data_as(_i64p), plan.part_row0.ctypes.data_as(_i64p), plan.sweep0.ctypes.data_as(_i64p),
                plan.nsteps.ctypes.data_as(_i32p), plan.hdr_off.ctypes.data_as(_i64p), plan.batch_off.ctypes.data_as(_i64p),
                plan.flush_off.ctypes.data_as(_i64p), plan.codes.ctypes.data_as(u32p), plan.vals.ctypes.data_as(_dp),
                plan.offs.ctypes.data_as(u16p), plan.flush_rows.ctypes.data_as(_i32p), plan.m, plan.ncols, n_chunks, _p(X), X.shape[0],
                _p(Y), Y.shape[0])
        return rc, Y

    def op_apply(self, rowptr, col, val, X):
        """A * X in the current partition mode (local rows; ghosts through the halo hook)."""
        X = _f(X)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        m = rowptr.size - 1
        Y = np.zeros((m, X.shape[1]), order="F")
        self.lib.orc_op_apply.argtypes = [C.c_int, _i64p, _i32p, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        self.lib.orc_op_apply(m, rowptr.ctypes.data_as(_i64p), col.ctypes.data_as(_i32p), _p(val), X.shape[1], _p(X), X.shape[0], _p(Y), m)
        return Y

    def find_largest(self, vals, N):
        vals = np.ascontiguousarray(vals, dtype=np.float64).ravel()
        out = np.zeros(N, dtype=np.int32)
        self.lib.orc_find_largest(_p(vals), vals.size, N, out.ctypes.data_as(_ip))
        return out

    def sb03md(self, A, Cm, trans="T"):
        """Solve A X + X A^T = scale*C (trans='T') as SLICOT SB03MD('C','X','N',trans) would."""
        A, X = _f(A).copy(order="F"), _f(Cm).copy(order="F")
        n = A.shape[0]
        scale = C.c_double(1.0)
        info = self.lib.orc_sb03md(trans.encode()[0:1], n, _p(A), n, _p(X), n, C.byref(scale))
        return X, scale.value, info

    def dense_solve(self, A, B):
        A, B = _f(A), _f(B)
        n = A.shape[0]
        X = np.zeros((n, n), order="F")
        info = self.lib.orc_dense_solve(n, _p(A), n, _p(B), n, _p(X), n)
        return X, info

    def dsyev(self, A):
        A = _f(A).copy(order="F")
        n = A.shape[0]
        w = np.zeros(n)
        info = self.lib.orc_dsyev(n, _p(A), n, _p(w))
        return w, A, info

    def resid_lanczos(self, AV, V, T, B, max_iter, rng_mode=1, seed=1, stream=0, row0=0):
        AV, V, T, B = _f(AV), _f(V), _f(T), _f(B)
        m, k = V.shape
        p = B.shape[1]
        H = np.zeros((max_iter + 1, max_iter + 1), order="F")
        ev = np.zeros(max_iter)
        evec = np.zeros((m, max_iter), order="F")
        Q = np.zeros((m, max_iter), order="F")
        steps = self.lib.orc_resid_lanczos(m, k, _p(AV), m, _p(V), m, _p(T), k, _p(B), m, p, max_iter, rng_mode, seed,
                                           stream, row0, _p(H), _p(ev), _p(evec), _p(Q))
        return dict(steps=steps, H=H, eigenvalues=ev[:steps].copy(), eigenvectors=evec[:, :steps].copy(order="F"),
                    Q=Q[:, :steps].copy(order="F"))

    def compute_restart_vectors(self, T, num, tol):
        T = _f(T)
        k = T.shape[0]
        X = np.zeros((k, k), order="F")
        kept = self.lib.orc_compute_restart_vectors(k, _p(T), k, num, tol, _p(X))
        return X[:, :kept].copy(order="F")

    def solve(self, A, B, params=None, M=None, V0=None, vcap=None, hist_cap=4096):
        """A (and M): dense ndarray or (rowptr, col, val) CSR triple.  Returns a dict."""
        B = _f(B)
        m, p = B.shape
        prm = params if isinstance(params, Params) else self.params(params)
        dense = None
        rp = ci = va = None
        if isinstance(A, tuple):
            rp = np.ascontiguousarray(A[0], dtype=np.int64)
            ci = np.ascontiguousarray(A[1], dtype=np.int32)
            va = np.ascontiguousarray(A[2], dtype=np.float64)
        else:
            dense = _f(A)
        mrp = mci = mva = None
        if M is not None:
            mrp = np.ascontiguousarray(M[0], dtype=np.int64)
            mci = np.ascontiguousarray(M[1], dtype=np.int32)
            mva = np.ascontiguousarray(M[2], dtype=np.float64)
        k0 = 0 if V0 is None else V0.shape[1]
        if vcap is None:
            base = prm.restart_size if prm.restart_size > 0 else 100
            vcap = max(k0, min(base, m))
            if prm.restart_size <= 0:
                vcap = min(m, max(vcap, 100) + 100 * 20)
        vcap = max(vcap, 1)
        V = np.zeros((m, vcap), order="F")
        if V0 is not None:
            V[:, :k0] = V0
        T = np.zeros((vcap, vcap), order="F")
        k = C.c_int(k0)
        trips = C.c_int(0)
        hist = np.zeros(hist_cap)
        null64 = C.cast(None, _i64p)
        null32 = C.cast(None, _i32p)
        nulld = C.cast(None, _dp)
        ret = self.lib.orc_solve(
            m, _p(dense) if dense is not None else nulld, m,
            rp.ctypes.data_as(_i64p) if rp is not None else null64,
            ci.ctypes.data_as(_i32p) if ci is not None else null32,
            _p(va) if va is not None else nulld,
            mrp.ctypes.data_as(_i64p) if mrp is not None else null64,
            mci.ctypes.data_as(_i32p) if mci is not None else null32,
            _p(mva) if mva is not None else nulld,
            _p(B), m, p, C.byref(prm), _p(V), m, vcap, C.byref(k), _p(T), vcap, _p(hist), hist_cap, C.byref(trips))
        kk = k.value
        ts = np.zeros(max(trips.value, 1))
        self.lib.orc_trip_seconds.restype = C.c_int
        self.lib.orc_trip_seconds.argtypes = [_dp, C.c_int]
        nt = self.lib.orc_trip_seconds(_p(ts), ts.size)
        return dict(ret=ret, V=V[:, :kk].copy(order="F"), T=T[:kk, :kk].copy(order="F"), trips=trips.value,
                    res_hist=hist[:min(trips.value, hist_cap)].copy(), trip_seconds=ts[:min(nt, ts.size)].copy())


class Reference:
    """The reference's own Stl sources compiled by oracle/Makefile (build container only)."""

    def __init__(self, path=REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path)
        L.ref_norm.restype = C.c_double
        L.ref_norm_inf.restype = C.c_double
        L.ref_srand.argtypes = [C.c_uint]
        L.ref_compute_restart_vectors.argtypes = [C.c_int, _dp, C.c_int, C.c_double, _dp]
        L.ref_resid_lanczos.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]

    def srand(self, s):
        self.lib.ref_srand(s)

    def random(self, m, n):
        X = np.zeros((m, n), order="F")
        self.lib.ref_random(m, n, _p(X))
        return X

    def dot(self, X, Y):
        X, Y = _f(X), _f(Y)
        Cm = np.zeros((X.shape[1], Y.shape[1]), order="F")
        self.lib.ref_dot(X.shape[0], X.shape[1], Y.shape[1], _p(X), _p(Y), _p(Cm))
        return Cm

    def mult(self, X, Cm):
        X, Cm = _f(X), _f(Cm)
        Y = np.zeros((X.shape[0], Cm.shape[1]), order="F")
        self.lib.ref_mult(X.shape[0], X.shape[1], Cm.shape[1], _p(X), _p(Cm), _p(Y))
        return Y

    def mult_t(self, X, Cm):
        X, Cm = _f(X), _f(Cm)
        Y = np.zeros((X.shape[1], Cm.shape[1]), order="F")
        self.lib.ref_mult_t(X.shape[0], X.shape[1], Cm.shape[1], _p(X), _p(Cm), _p(Y))
        return Y

    def norm(self, X):
        X = _f(X)
        return self.lib.ref_norm(X.shape[0], X.shape[1], _p(X))

    def norm_inf(self, X):
        X = _f(X)
        return self.lib.ref_norm_inf(X.shape[0], X.shape[1], _p(X))

    def orthogonalize(self, V1, V2=None):
        V1 = _f(V1)
        m, n1 = V1.shape
        n2 = 0 if V2 is None else _f(V2).shape[1]
        V2 = _f(V2) if V2 is not None else np.zeros((m, 1), order="F")
        out = np.zeros((m, n1 + n2), order="F")
        self.lib.ref_orthogonalize(m, n1, _p(V1), n2, _p(V2), _p(out))
        return out

    def eigs(self, A):
        A = _f(A)
        n = A.shape[0]
        V = np.zeros((n, n), order="F")
        d = np.zeros(n)
        info = self.lib.ref_eigs(n, _p(A), _p(V), _p(d))
        return d, V, info

    def find_largest(self, vals, N):
        vals = np.ascontiguousarray(vals, dtype=np.float64).ravel()
        out = np.zeros(N, dtype=np.int32)
        self.lib.ref_find_largest(_p(vals), vals.size, N, out.ctypes.data_as(_ip))
        return out

    def resid_lanczos(self, AV, V, T, B, max_iter):
        AV, V, T, B = _f(AV), _f(V), _f(T), _f(B)
        m, k = V.shape
        p = B.shape[1]
        H = np.zeros((max_iter + 1, max_iter + 1), order="F")
        ev = np.zeros(max_iter)
        evec = np.zeros((m, max_iter), order="F")
        steps = self.lib.ref_resid_lanczos(m, k, p, _p(AV), _p(V), _p(T), _p(B), max_iter, _p(H), _p(ev), _p(evec))
        return dict(steps=steps, H=H, eigenvalues=ev[:steps].copy(), eigenvectors=evec[:, :steps].copy(order="F"))

    def compute_restart_vectors(self, T, num, tol):
        T = _f(T)
        k = T.shape[0]
        X = np.zeros((k, k), order="F")
        kept = self.lib.ref_compute_restart_vectors(k, _p(T), num, tol, _p(X))
        return X[:, :kept].copy(order="F")
