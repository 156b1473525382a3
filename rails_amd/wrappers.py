"""Python mirror of the wrapper pair of the hot path, over the C ABI of librails_hip.so.

Names and argument meaning follow the reference's backend contract
(src/StlWrapper.hpp:32-90, src/Epetra_OperatorWrapper.hpp): `HipOperatorWrapper` plays the
Matrix role (`A * X`, `A.transpose() * X`), `HipMultiVectorWrapper` the MultiVector role
(`dot`, `norm`, `orthogonalize`, `view`, `push_back`, `random`, `* DenseMatrix`, ...).  Dense
results come back as numpy arrays (column-major), the DenseMatrix role staying on the host.

The C++ header-only wrappers in rails_amd/include/rails/ are the drop-in for the reference's
templated Solver; these Python classes exist so the parity tests read like the reference's tests.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import check

_dp = C.POINTER(C.c_double)


def _f(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


def _p(a):
    return a.ctypes.data_as(_dp)


class Context:
    """Device context: device ordinal, stream, RNG seed, row partition, all-reduce hook."""

    def __init__(self, device=0, stream=None, seed=1, first_stream=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.rails_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)), "rails_ctx_create")
        self.h = h
        self._cb = None
        self._solvers = weakref.WeakSet()  # closed before the context (their C++ side owns panels of this context)
        self.set_seed(seed, first_stream)

    def set_seed(self, seed, first_stream=0):
        check(self.lib.rails_ctx_set_seed(self.h, seed, first_stream), "rails_ctx_set_seed")

    def set_partition(self, rank, nranks, row0, m_global):
        check(self.lib.rails_ctx_set_partition(self.h, rank, nranks, row0, m_global), "rails_ctx_set_partition")

    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, n:int, stream:int) -> 0 on success."""
        def tramp(user, buf, n, stream):
            try:
                return int(pyfunc(buf or 0, n, stream or 0) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.ALLREDUCE_FN(tramp)
        check(self.lib.rails_ctx_set_allreduce(self.h, self._cb, None), "rails_ctx_set_allreduce")

    @staticmethod
    def rccl_unique_id():
        """128 bytes made by ONE rank and handed to every rank of the partition (any transport) for init_rccl."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        check(lib.rails_rccl_unique_id(buf), "rails_rccl_unique_id")
        return buf.raw

    def init_rccl(self, unique_id, nranks, rank):
        """The library's own RCCL communicator (collective: every rank calls it): all-reduces and ghost-row exchanges then run
        inside the library on the context's stream, no hook needed."""
        assert len(unique_id) == 128
        check(self.lib.rails_ctx_init_rccl(self.h, C.c_char_p(unique_id), nranks, rank), "rails_ctx_init_rccl")

    def set_rccl(self, comm):
        """hand over an existing ncclComm_t (an integer address), or None to drop the communicator the context holds"""
        check(self.lib.rails_ctx_set_rccl(self.h, C.c_void_p(comm) if comm else None), "rails_ctx_set_rccl")

    def rccl_size(self):
        return self.lib.rails_ctx_rccl_size(self.h)

    def sync(self):
        check(self.lib.rails_ctx_sync(self.h), "rails_ctx_sync")

    def stream(self):
        return self.lib.rails_ctx_stream(self.h)

    def stats(self):
        import json
        buf = C.create_string_buffer(1024)
        check(self.lib.rails_ctx_stats(self.h, buf, 1024), "rails_ctx_stats")
        return json.loads(buf.value.decode())

    def enable_library_gemm(self):
        """opt-in (with RAILS_WIDE_GEMM=rocblas in the environment): the basis rotation of restarts through rocBLAS instead of the library's own kernel"""
        check(self.lib.rails_ctx_enable_library_gemm(self.h), "rails_ctx_enable_library_gemm")

    def set_meter(self, on=True):
        """device-busy meter: stats()["gpu_busy_ms"] adds up the time the GPU worked for this context"""
        check(self.lib.rails_ctx_set_meter(self.h, 1 if on else 0), "rails_ctx_set_meter")

    def timer_start(self):
        check(self.lib.rails_timer_start(self.h), "rails_timer_start")

    def timer_stop(self):
        ms = C.c_double(0.0)
        check(self.lib.rails_timer_stop(self.h, C.byref(ms)), "rails_timer_stop")
        return ms.value

    def close(self):
        if self.h:
            for s in list(self._solvers):
                s.close()
            self.lib.rails_lanczos_release(self.h)
            self.lib.rails_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Panel:
    """Owning handle of a device panel (shared by views, like the shared_ptr in StlWrapper.hpp:13)."""

    def __init__(self, ctx, m, capacity):
        self.ctx = ctx
        h = C.c_void_p()
        m = int(m)
        check(ctx.lib.rails_panel_create(ctx.h, m, int(capacity), C.byref(h)), "rails_panel_create")
        self.h = h
        self.m = m

    @property
    def capacity(self):
        return self.ctx.lib.rails_panel_capacity(self.h)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.rails_panel_destroy(self.h)
        except Exception:
            pass


class _BorrowedPanel:
    """a panel handle owned by the library (or by another object): same attributes as _Panel, no destructor"""

    def __init__(self, ctx, h, m):
        self.ctx, self.h, self.m = ctx, C.c_void_p(h) if not isinstance(h, C.c_void_p) else h, m

    @property
    def capacity(self):
        return self.ctx.lib.rails_panel_capacity(self.h)


class HipMultiVectorWrapper:
    """Row-partitioned device multivector: a column window [c0, c0+n) of a shared panel."""

    def __init__(self, ctx, m=None, n=0, capacity=None, data=None):
        self.ctx = ctx
        if data is not None:
            data = _f(data)
            m, n = data.shape
        cap = max(capacity or 0, n, 1)
        self.panel = _Panel(ctx, m, cap)
        self.c0 = 0
        self.n = n
        self.is_view = False
        self.orthogonalized = 0
        if data is not None and n:
            check(ctx.lib.rails_panel_upload(ctx.h, self.panel.h, 0, n, _p(data), data.shape[0]), "rails_panel_upload")

    @classmethod
    def _borrow(cls, ctx, panel_ptr, c0, n, m):
        """a view onto a panel the library owns (operator callbacks): nothing is freed when it goes away"""
        v = cls.__new__(cls)
        v.ctx = ctx
        v.panel = _BorrowedPanel(ctx, panel_ptr, m)
        v.c0, v.n, v.is_view, v.orthogonalized = c0, n, True, 0
        return v

    # --- shape -------------------------------------------------------------------------------
    def M(self):
        return self.panel.m

    def N(self):
        return self.n

    def _alias(self, c0, n, view):
        o = object.__new__(HipMultiVectorWrapper)
        o.ctx, o.panel, o.c0, o.n, o.is_view, o.orthogonalized = self.ctx, self.panel, c0, n, view, 0
        return o

    def view(self, a=-1, b=-1):
        """view(i) = column i; view(a, b) = columns a..b inclusive; view() = all (src/StlWrapper.cpp:323-340)."""
        if a < 0:
            return self._alias(self.c0, self.n, True)
        num = (b - a + 1) if b > 0 else 1
        return self._alias(self.c0 + a, num, True)

    def resize(self, n):
        """Capacity-preserving resize (src/StlWrapper.cpp:225-263)."""
        self.orthogonalized = min(self.orthogonalized, n)
        if self.c0 + n > self.panel.capacity:
            check(self.ctx.lib.rails_panel_reserve(self.ctx.h, self.panel.h, self.c0 + n), "rails_panel_reserve")
        self.n = n

    def copy(self):
        o = HipMultiVectorWrapper(self.ctx, self.M(), self.n, capacity=max(self.n, 1))
        if self.n:
            check(self.ctx.lib.rails_panel_copy(self.ctx.h, self.panel.h, self.c0, self.n, o.panel.h, 0), "rails_panel_copy")
        o.orthogonalized = self.orthogonalized
        return o

    def assign(self, other):
        """operator=: a non-view target shares storage, a view target copies in (src/StlWrapper.cpp:65-121)."""
        if isinstance(other, (int, float)):
            check(self.ctx.lib.rails_panel_fill(self.ctx.h, self.panel.h, self.c0, self.n, float(other)), "rails_panel_fill")
            self.orthogonalized = 0
            return self
        if not self.is_view:
            self.panel, self.c0, self.n, self.orthogonalized = other.panel, other.c0, other.n, other.orthogonalized
            return self
        assert self.n == other.n
        check(self.ctx.lib.rails_panel_copy(self.ctx.h, other.panel.h, other.c0, self.n, self.panel.h, self.c0), "rails_panel_copy")
        return self

    def push_back(self, other):
        n = self.n
        self.resize(n + other.n)
        check(self.ctx.lib.rails_panel_copy(self.ctx.h, other.panel.h, other.c0, other.n, self.panel.h, self.c0 + n), "rails_panel_copy")

    # --- data --------------------------------------------------------------------------------
    def to_host(self):
        out = np.zeros((self.M(), self.n), order="F")
        if self.n:
            check(self.ctx.lib.rails_panel_download(self.ctx.h, self.panel.h, self.c0, self.n, _p(out), max(1, self.M())), "rails_panel_download")
        return out

    def from_host(self, data):
        data = _f(data)
        assert data.shape == (self.M(), self.n)
        check(self.ctx.lib.rails_panel_upload(self.ctx.h, self.panel.h, self.c0, self.n, _p(data), data.shape[0]), "rails_panel_upload")
        self.orthogonalized = 0

    def random(self):
        check(self.ctx.lib.rails_panel_random(self.ctx.h, self.panel.h, self.c0, self.n), "rails_panel_random")
        self.orthogonalized = 0

    # --- BLAS-1 ------------------------------------------------------------------------------
    def __imul__(self, s):
        check(self.ctx.lib.rails_panel_scale(self.ctx.h, self.panel.h, self.c0, self.n, float(s)), "rails_panel_scale")
        self.orthogonalized = 0
        return self

    def __itruediv__(self, s):
        return self.__imul__(1.0 / s)  # src/StlWrapper.cpp:139-143

    def __iadd__(self, other):
        check(self.ctx.lib.rails_panel_axpy(self.ctx.h, 1.0, other.panel.h, other.c0, self.n, self.panel.h, self.c0), "rails_panel_axpy")
        self.orthogonalized = 0
        return self

    def __isub__(self, other):
        check(self.ctx.lib.rails_panel_axpy(self.ctx.h, -1.0, other.panel.h, other.c0, self.n, self.panel.h, self.c0), "rails_panel_axpy")
        self.orthogonalized = 0
        return self

    def __rmul__(self, s):
        o = self.copy()
        o *= s
        return o

    # --- reductions / products ---------------------------------------------------------------
    def dot(self, other):
        """X^T Y as a host (numpy, column-major) dense matrix (src/StlWrapper.cpp:394-412)."""
        out = np.zeros((self.n, other.n), order="F")
        check(self.ctx.lib.rails_gram(self.ctx.h, self.panel.h, self.c0, self.n, other.panel.h, other.c0, other.n, _p(out), max(1, self.n)),
              "rails_gram")
        return out

    def norm(self):
        """Spectral 2-norm: sqrt(max |eig(X^T X)|) (src/StlWrapper.cpp:265-289)."""
        if self.n == 0:
            return 0.0
        G = self.dot(self)
        w = np.zeros(self.n)
        info = C.c_int(0)
        self.ctx.lib.rails_dsyev(b"V", b"U", self.n, _p(G), self.n, _p(w), C.byref(info))
        if info.value:
            raise _lib.RailsError("rails_dsyev info = %d" % info.value)
        return float(np.sqrt(np.abs(w)).max())

    def matmul(self, Cm):
        """self * DenseMatrix (src/StlWrapper.cpp:168-187)."""
        Cm = _f(Cm)
        assert Cm.shape[0] == self.n
        o = HipMultiVectorWrapper(self.ctx, self.M(), Cm.shape[1], capacity=max(1, Cm.shape[1]))
        check(self.ctx.lib.rails_panel_gemm(self.ctx.h, 1.0, self.panel.h, self.c0, self.n, _p(Cm), max(1, Cm.shape[0]), Cm.shape[1], 0.0,
                                            o.panel.h, 0), "rails_panel_gemm")
        return o

    def gemm_into(self, Cm, out, alpha=1.0, beta=0.0):
        Cm = _f(Cm)
        check(self.ctx.lib.rails_panel_gemm(self.ctx.h, alpha, self.panel.h, self.c0, self.n, _p(Cm), max(1, Cm.shape[0]), Cm.shape[1], beta,
                                            out.panel.h, out.c0), "rails_panel_gemm")

    def orthogonalize(self, method=0):
        used = C.c_int(0)
        k0 = self.orthogonalized
        if self.c0 != 0:
            raise _lib.RailsError("orthogonalize on a view that does not start at column 0 is not supported")
        check(self.ctx.lib.rails_orthogonalize(self.ctx.h, self.panel.h, k0, self.n - k0, method, C.byref(used)), "rails_orthogonalize")
        self.orthogonalized = self.n
        return used.value


class HipOperatorWrapper:
    """Device CSR operator: the Matrix role (`A * X`, `A.transpose() * X`)."""

    def __init__(self, ctx, rowptr, col, val, ncols_ext=None, _handle=None, _trans=False):
        self.ctx = ctx
        self.trans = _trans
        if _handle is not None:
            self.h = _handle
            return
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        m = rowptr.size - 1
        h = C.c_void_p()
        check(ctx.lib.rails_csr_create(ctx.h, m, ncols_ext if ncols_ext is not None else m, rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                       col.ctypes.data_as(C.POINTER(C.c_int32)), _p(val), C.byref(h)), "rails_csr_create")
        self.h = _Handle(ctx, h)

    @classmethod
    def rect(cls, ctx, n_rows, n_cols, rowptr, col, val):
        """A rectangular operator (include/rails_hip.h: rails_csr_create_rect): X of a product has n_cols rows, Y n_rows."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        assert rowptr.size == n_rows + 1
        h = C.c_void_p()
        check(ctx.lib.rails_csr_create_rect(ctx.h, n_rows, n_cols, rowptr.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                                            _p(val), C.byref(h)), "rails_csr_create_rect")
        op = cls(ctx, None, None, None, _handle=_Handle(ctx, h))
        op.n_rows, op.n_cols = n_rows, n_cols
        return op

    @classmethod
    def from_callback(cls, ctx, m, pyfunc):
        """An operator given by its action (include/rails_hip.h: rails_csr_create_callback): pyfunc(trans, X, Y) receives two
        HipMultiVectorWrapper views (m x nc windows of the library's panels) and must set Y = op(A) X."""
        def tramp(user, trans, xp, xc0, nc, yp, yc0):
            try:
                X = HipMultiVectorWrapper._borrow(ctx, xp, xc0, nc, m)
                Y = HipMultiVectorWrapper._borrow(ctx, yp, yc0, nc, m)
                return int(pyfunc(bool(trans), X, Y) or 0)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = _lib.APPLY_FN(tramp)
        h = C.c_void_p()
        check(ctx.lib.rails_csr_create_callback(ctx.h, m, cb, None, C.byref(h)), "rails_csr_create_callback")
        op = cls(ctx, None, None, None, _handle=_Handle(ctx, h))
        op._apply_cb = cb  # keep the trampoline alive as long as the operator
        op.h._keep = cb
        return op

    def M(self):
        return self.ctx.lib.rails_csr_rows(self.h.h)

    def N(self):
        return self.ctx.lib.rails_csr_rows(self.h.h)

    def nnz(self):
        return self.ctx.lib.rails_csr_nnz(self.h.h)

    def transpose(self):
        return HipOperatorWrapper(self.ctx, None, None, None, _handle=self.h, _trans=not self.trans)

    def set_variant(self, v):
        check(self.ctx.lib.rails_csr_set_variant(self.h.h, v), "rails_csr_set_variant")

    def last_kernel(self):
        return self.ctx.lib.rails_csr_last_kernel(self.h.h).decode()

    def prepare(self, nc):
        """Set-up for many products of nc columns (rails_csr_prepare): builds the sweep kernel's schedule now where it applies instead of
        after the 16th such product.  Returns True when the sweep kernel will take them."""
        ready = C.c_int(0)
        check(self.ctx.lib.rails_csr_prepare(self.ctx.h, self.h.h, 1 if self.trans else 0, nc, C.byref(ready)), "rails_csr_prepare")
        return bool(ready.value)

    def sweep_stats(self, nc):
        """Schedule statistics of the sweep kernel for nc columns (zeros until the schedule exists: prepare(), or the 16th product)."""
        out = (C.c_double * 4)()
        check(self.ctx.lib.rails_csr_sweep_stats(self.h.h, nc, out), "rails_csr_sweep_stats")
        return {"efficiency": out[0], "staged_rows_per_row": out[1], "trips": int(out[2]), "built": bool(out[3])}

    def set_halo(self, plan, pyfunc):
        """Install the ghost-row plan (rails_amd.partition.HaloPlan) and hook pyfunc(send_ptr, recv_ptr, ncols, stream)."""
        def tramp(user, send, recv, nc, stream):
            try:
                return int(pyfunc(send or 0, recv or 0, nc, stream or 0) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._halo_cb = _lib.HALO_FN(tramp) if pyfunc is not None else _lib.HALO_FN(0)  # None: the library's RCCL exchange
        rows = np.ascontiguousarray(plan.send_rows, dtype=np.int64)
        check(self.ctx.lib.rails_csr_set_halo(self.h.h, plan.n_send, rows.ctypes.data_as(C.POINTER(C.c_int64)), plan.n_ghost,
                                              self._halo_cb, None), "rails_csr_set_halo")
        sc = np.ascontiguousarray(plan.send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(plan.recv_counts, dtype=np.int64)
        check(self.ctx.lib.rails_csr_set_halo_counts(self.h.h, plan.nranks, sc.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     rc.ctypes.data_as(C.POINTER(C.c_int64))), "rails_csr_set_halo_counts")

    def apply(self, X, Y=None):
        """Y = op(A) * X (src/LyapunovSolver.hpp:146)."""
        if Y is None:
            Y = HipMultiVectorWrapper(self.ctx, getattr(self, "n_rows", X.M()), X.n, capacity=max(1, X.n))
        check(self.ctx.lib.rails_spmm(self.ctx.h, self.h.h, 1 if self.trans else 0, X.panel.h, X.c0, X.n, Y.panel.h, Y.c0), "rails_spmm")
        return Y

    def __mul__(self, X):
        return self.apply(X)


class _Handle:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.rails_csr_destroy(self.h)
        except Exception:
            pass


def resid_lanczos(ctx, AV, V, T, B, max_iter, MV=None):
    """Fused residual Lanczos (src/LyapunovSolver.hpp:367-447).  Returns dict(steps, H, eigenvalues, v)."""
    T = _f(T)
    k = AV.n
    H = np.zeros((max_iter + 1, max_iter + 1), order="F")
    steps = C.c_int(0)
    MVp = MV if MV is not None else V
    check(ctx.lib.rails_resid_lanczos(ctx.h, AV.panel.h, AV.c0, MVp.panel.h, MVp.c0, k, _p(T), max(1, k), B.panel.h, B.c0, B.n, max_iter,
                                      _p(H), max_iter + 1, C.byref(steps)), "rails_resid_lanczos")
    s = steps.value
    Hs = np.asfortranarray(H[:s, :s].copy())
    w = np.zeros(s)
    info = C.c_int(0)
    ctx.lib.rails_dsyev(b"V", b"U", s, _p(Hs), s, _p(w), C.byref(info))
    return dict(steps=s, H=H, eigenvalues=w, v=Hs)


def lanczos_vectors(ctx, S, out):
    S = _f(S)
    check(ctx.lib.rails_lanczos_vectors(ctx.h, _p(S), S.shape[0], S.shape[1], out.panel.h, out.c0), "rails_lanczos_vectors")
