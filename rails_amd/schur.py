"""Schur-complement operator for descriptor systems with a singular diagonal mass matrix.

The reference builds it with Trilinos (src/SchurOperator.cpp:51-214): rows / columns are split into the set 1 where the mass
matrix is zero (algebraic constraints) and the set 2 where it is not; the operator the Lyapunov solver sees is

    S = A22 - A21 * A11^-1 * A12                                                   (src/SchurOperator.cpp:181-214)

with a sparse LU factorisation of A11 (Amesos KLU there, SuperLU through scipy here, on the host like the reference's serial KLU).
Here `S * X`: A22 * X, A12 * X and A21 * Z are CSR SpMM kernels on the device (A12, A21 as rectangular operators, rails_csr_create_rect):
X never leaves the device; what crosses PCIe per product is the m1 x nc block A12 X on its way to the LU solve and the solution Z on
its way back (m1 = number of algebraic constraints).  The operator plugs into the solver through
the C ABI's operator-callback handle (rails_csr_create_callback), so both back ends of the solver template run on it unchanged.
Single rank, like the reference ("TODO: Fix these maps to work in parallel runs", src/SchurOperator.cpp:226).
"""
import numpy as np

from ._lib import check
from .wrappers import HipMultiVectorWrapper, HipOperatorWrapper


class SchurOperator:
    def __init__(self, ctx, A, mass_diagonal, tol=1e-15):
        """A: (rowptr, col, val) CSR of the full n x n operator; mass_diagonal: the n diagonal entries of M."""
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
        self.op = HipOperatorWrapper.from_callback(ctx, self.m2, self._apply)

    def _workspace(self, nc):
        """the three panels a product needs (A12 X and the solution of the A11 system on set 1, A21 Z on set 2), kept per width: a product
        allocates nothing after the first of its width"""
        ws = self._ws.get(nc)
        if ws is None:
            ws = (HipMultiVectorWrapper(self.ctx, self.m1, nc, capacity=max(1, nc)), HipMultiVectorWrapper(self.ctx, self.m1, nc, capacity=max(1, nc)),
                  HipMultiVectorWrapper(self.ctx, self.m2, nc, capacity=max(1, nc)))
            self._ws[nc] = ws
        return ws

    def _apply(self, trans, X, Y):
        """Y = S X (or S' X): three device SpMMs; the LU solve with A11 on the host, on an m1 x nc block.  The only synchronisation is
        the one the host solve needs (the block has to have arrived); everything after it is queued and returns."""
        self.applies += X.n
        lib = self.ctx.lib
        W, Zd, tmp = self._workspace(X.n)
        check(lib.rails_spmm(self.ctx.h, self.A22.h.h, 1 if trans else 0, X.panel.h, X.c0, X.n, Y.panel.h, Y.c0), "rails_spmm")
        first, second = (self.dA12, self.dA21) if not trans else (self.dA21t, self.dA12t)  # S' = A22' - A12' A11^-T A21'
        check(lib.rails_spmm(self.ctx.h, first.h.h, 0, X.panel.h, X.c0, X.n, W.panel.h, 0), "rails_spmm")
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
