"""Schur-complement operator for descriptor systems with a singular diagonal mass matrix.

The reference builds it with Trilinos (src/SchurOperator.cpp:51-214): rows / columns are split into the set 1 where the mass
matrix is zero (algebraic constraints) and the set 2 where it is not; the operator the Lyapunov solver sees is

    S = A22 - A21 * A11^-1 * A12                                                   (src/SchurOperator.cpp:181-214)

with a sparse LU factorisation of A11 (Amesos KLU there, SuperLU through scipy here: computed once, on the host, like the reference's).
Here `S * X`: A22 * X, A12 * X and A21 * Z are CSR SpMM kernels on the device (A12, A21 as rectangular operators, rails_csr_create_rect)
and the solve with A11 applies the LU factors on the device as well (DeviceLU below: level-scheduled sparse triangular solves,
rails_amd/csrc/sptrsv.hip) -- nothing of a product crosses PCIe and nothing in it synchronises.  (`device_solve=False` /
RAILS_SCHUR_HOST_SOLVE=1: the solve on the host, as the reference does it inside Apply; then the m1 x nc block A12 X and the solution Z
cross PCIe, m1 = number of algebraic constraints.)  The operator plugs into the solver through
the C ABI's operator-callback handle (rails_csr_create_callback), so both back ends of the solver template run on it unchanged.
Single rank, like the reference ("TODO: Fix these maps to work in parallel runs", src/SchurOperator.cpp:226).
"""
import ctypes as C

import numpy as np

from ._lib import check
from .wrappers import HipMultiVectorWrapper, HipOperatorWrapper


class DeviceLU:
    """The factors of a host LU factorisation (scipy's SuperLU object: Pr A Pc = L U, L unit lower) applied on the device:
    A^-1 b = Pc U^-1 L^-1 Pr b and A^-T b = Pr' L^-T U^-T Pc' b, each a row permutation, two level-scheduled triangular solves
    (rails_sptrsv_solve) and a row permutation -- panels in, panels out, nothing crosses PCIe (src/SchurOperator.cpp:193-200 is the
    reference's host-side solve inside Apply)."""

    def __init__(self, ctx, lu):
        import scipy.sparse as sp

        self.ctx, self.n = ctx, lu.shape[0]
        lib = ctx.lib
        self._tri, self._idx = {}, {}

        def tri(name, M, lower, unit):
            M = sp.csr_matrix(M)
            if unit:  # SuperLU stores the ones of L's diagonal
                M = M - sp.identity(self.n, format="csr")
                M.eliminate_zeros()
            M.sort_indices()
            h = C.c_void_p()
            rp, ci, va = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
            check(lib.rails_sptrsv_create(ctx.h, self.n, rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                          va.ctypes.data_as(C.POINTER(C.c_double)), 1 if lower else 0, 1 if unit else 0, C.byref(h)), "rails_sptrsv_create")
            self._tri[name] = h

        tri("L", lu.L, True, True)
        tri("U", lu.U, False, False)
        tri("Lt", lu.L.T, False, True)
        tri("Ut", lu.U.T, True, False)
        for name, perm in (("r", lu.perm_r), ("c", lu.perm_c)):
            h = C.c_void_p()
            p32 = np.ascontiguousarray(perm, dtype=np.int32)
            check(lib.rails_index_upload(ctx.h, p32.ctypes.data_as(C.POINTER(C.c_int32)), self.n, C.byref(h)), "rails_index_upload")
            self._idx[name] = h

    def levels(self):
        return {k: int(self.ctx.lib.rails_sptrsv_levels(h)) for k, h in self._tri.items()}

    def solve(self, B, tmp, out, trans=False):
        """out = A^-1 B (or A^-T B): B, tmp, out are three distinct panels of n rows and the same width; B is left as it was"""
        lib, h = self.ctx.lib, self.ctx.h
        nc = B.n
        assert tmp.n == nc and out.n == nc
        if not trans:
            # (Pr b)[perm_r[i]] = b[i];  x[i] = y[perm_c[i]]
            check(lib.rails_panel_permute_rows(h, B.panel.h, B.c0, nc, self._idx["r"], 1, tmp.panel.h, tmp.c0), "rails_panel_permute_rows")
            check(lib.rails_sptrsv_solve(h, self._tri["L"], tmp.panel.h, tmp.c0, nc), "rails_sptrsv_solve")
            check(lib.rails_sptrsv_solve(h, self._tri["U"], tmp.panel.h, tmp.c0, nc), "rails_sptrsv_solve")
            check(lib.rails_panel_permute_rows(h, tmp.panel.h, tmp.c0, nc, self._idx["c"], 0, out.panel.h, out.c0), "rails_panel_permute_rows")
        else:
            # A' = Pc U' L' Pr:  (Pc' b)[perm_c[i]] = b[i];  x[i] = y[perm_r[i]]
            check(lib.rails_panel_permute_rows(h, B.panel.h, B.c0, nc, self._idx["c"], 1, tmp.panel.h, tmp.c0), "rails_panel_permute_rows")
            check(lib.rails_sptrsv_solve(h, self._tri["Ut"], tmp.panel.h, tmp.c0, nc), "rails_sptrsv_solve")
            check(lib.rails_sptrsv_solve(h, self._tri["Lt"], tmp.panel.h, tmp.c0, nc), "rails_sptrsv_solve")
            check(lib.rails_panel_permute_rows(h, tmp.panel.h, tmp.c0, nc, self._idx["r"], 0, out.panel.h, out.c0), "rails_panel_permute_rows")

    def close(self):
        for hnd in self._tri.values():
            self.ctx.lib.rails_sptrsv_destroy(hnd)
        for hnd in self._idx.values():
            self.ctx.lib.rails_index_free(self.ctx.h, hnd)
        self._tri, self._idx = {}, {}


class SchurOperator:
    def __init__(self, ctx, A, mass_diagonal, tol=1e-15, device_solve=None):
        """A: (rowptr, col, val) CSR of the full n x n operator; mass_diagonal: the n diagonal entries of M.  device_solve: apply the LU
        factors of A11 on the device (DeviceLU; the default, RAILS_SCHUR_HOST_SOLVE=1 turns it off) or solve on the host inside every
        product as the reference does."""
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla

        rowptr, col, val = A
        n = rowptr.size - 1
        d = np.asarray(mass_diagonal, dtype=np.float64)
        assert d.shape == (n,)
        self.idx1 = np.flatnonzero(np.abs(d) < tol)   # M_ii = 0 (src/SchurOperator.cpp:70-76)
        self.idx2 = np.flatnonzero(np.abs(d) >= tol)
        self.m1, self.m2 = self.idx1.size, self.idx2.size
        if self.m1 == 0:
            raise ValueError("the mass matrix is nonsingular: no Schur complement to take")
        As = sp.csr_matrix((val, col, rowptr), shape=(n, n))
        self.A11 = As[self.idx1][:, self.idx1].tocsc()
        self.A12 = As[self.idx1][:, self.idx2].tocsr()
        self.A21 = As[self.idx2][:, self.idx1].tocsr()
        A22 = As[self.idx2][:, self.idx2].tocsr()
        A22.sort_indices()
        self.lu = spla.splu(self.A11)  # symbolic + numeric factorisation (src/SchurOperator.cpp:171-176)
        self.mass22 = d[self.idx2].copy()
        self.ctx = ctx
        self.A22 = HipOperatorWrapper(ctx, A22.indptr.astype(np.int64), A22.indices.astype(np.int32), A22.data.astype(np.float64))

        def rect(Mx):
            Mx = Mx.tocsr()
            Mx.sort_indices()
            return HipOperatorWrapper.rect(ctx, Mx.shape[0], Mx.shape[1], Mx.indptr.astype(np.int64), Mx.indices.astype(np.int32), Mx.data.astype(np.float64))

        # the off-diagonal blocks and their transposes as device operators (a rectangular operator has no transposed apply)
        self.dA12, self.dA21 = rect(self.A12), rect(self.A21)
        self.dA12t, self.dA21t = rect(self.A12.T), rect(self.A21.T)
        self.applies = 0  # matrix-vector products, as SchurOperator::GetMVPs counts them
        self.host_bytes = 0  # bytes that crossed PCIe in _apply (diagnostics)
        self._ws = {}
        import os

        if device_solve is None:
            device_solve = os.environ.get("RAILS_SCHUR_HOST_SOLVE", "0") in ("", "0")
        self.dlu = DeviceLU(ctx, self.lu) if device_solve else None
        self.op = HipOperatorWrapper.from_callback(ctx, self.m2, self._apply)

    def _workspace(self, nc):
        """the panels a product needs (on set 1: A12 X, the permuted right-hand side the triangular solves work on, the solution of the A11
        system; on set 2: A21 Z), kept per width: a product allocates nothing after the first of its width"""
        ws = self._ws.get(nc)
        if ws is None:
            mk = lambda rows: HipMultiVectorWrapper(self.ctx, rows, nc, capacity=max(1, nc))
            ws = (mk(self.m1), mk(self.m1), mk(self.m1), mk(self.m2))
            self._ws[nc] = ws
        return ws

    def _apply(self, trans, X, Y):
        """Y = S X (or S' X): three device SpMMs around the solve with A11 -- on the device (two row permutations and two triangular solves:
        nothing leaves it and nothing synchronises), or on the host on an m1 x nc block (then the only synchronisation is the one the host
        solve needs: the block has to have arrived)."""
        self.applies += X.n
        lib = self.ctx.lib
        W, T1, Zd, tmp = self._workspace(X.n)
        check(lib.rails_spmm(self.ctx.h, self.A22.h.h, 1 if trans else 0, X.panel.h, X.c0, X.n, Y.panel.h, Y.c0), "rails_spmm")
        first, second = (self.dA12, self.dA21) if not trans else (self.dA21t, self.dA12t)  # S' = A22' - A12' A11^-T A21'
        check(lib.rails_spmm(self.ctx.h, first.h.h, 0, X.panel.h, X.c0, X.n, W.panel.h, 0), "rails_spmm")
        if self.dlu is not None:
            self.dlu.solve(W, T1, Zd, trans=bool(trans))
        else:
            Wh = W.to_host()
            Z = self.lu.solve(np.ascontiguousarray(Wh), trans="T" if trans else "N")
            self.host_bytes += 2 * Wh.nbytes
            Zd.from_host(np.asfortranarray(Z.reshape(self.m1, X.n)))
        check(lib.rails_spmm(self.ctx.h, second.h.h, 0, Zd.panel.h, 0, X.n, tmp.panel.h, 0), "rails_spmm")
        check(lib.rails_panel_axpy(self.ctx.h, -1.0, tmp.panel.h, 0, X.n, Y.panel.h, Y.c0), "rails_panel_axpy")
        return 0

    def restrict(self, B):
        """B on the set-2 unknowns: B2 - A21 A11^-1 B1 (matlab/RAILSschur.m:47-53,71-73; the reference's C++ driver imports the rows of
        B into the Schur operator's range map, src/main.cpp:83-88, which is the same thing when B vanishes on set 1)"""
        B = np.asarray(B, dtype=np.float64)
        B2 = B[self.idx2]
        if self.m1 and np.abs(B[self.idx1]).max() > np.sqrt(np.finfo(float).eps):
            B2 = B2 - self.A21 @ self.lu.solve(np.ascontiguousarray(B[self.idx1]))
        return np.asfortranarray(B2)

    def prolongate(self, V):
        """the solution on all unknowns from the solution on set 2: x1 = -A11^-1 A12 x2 (matlab/RAILSschur.m:75-77); X = Vf T Vf'"""
        V = np.asarray(V, dtype=np.float64)
        out = np.zeros((self.m1 + self.m2, V.shape[1]), order="F")
        out[self.idx2] = V
        out[self.idx1] = -self.lu.solve(np.ascontiguousarray(self.A12 @ V))
        return out

    def dense(self):
        """the Schur complement as a dense matrix (tests, small problems)"""
        A22 = self.A22.apply(HipMultiVectorWrapper(self.ctx, data=np.eye(self.m2))).to_host()
        return A22 - self.A21 @ self.lu.solve(self.A12.toarray())
