"""The hot path at BASELINE.json's full sizes (configs[2]: m = 1M rows, 27 nnz/row banded-random, B m x 16, Restart size 200 /
Reduced size 128 / Expand size 16 / Lanczos iterations 20; SpMM at 128 columns), checked through properties that do not need
a full-size CPU solve:

  SpMM    the oracle's CSR product on the same matrix at 16 columns (it finishes in about a second) and on sampled rows at 128
          columns; linearity A(aX + bZ) = a AX + b AZ; the adjoint identity <Z, A X> = <A' Z, X>; A * ones = row sums; the same on
          one 1M-row slab of configs[3]'s 27-point stencil (the LDS-staged kernel)
  Gram / orthogonalize / panel update   V'V = I after orthogonalize at 1M x 200; Q (Q'X) reproduces X for X in span(Q)
  solver  on both back ends: V'V = I, T = T', the reference's convergence criterion ||R||_2 < tol * ||B'B||_2
          (src/LyapunovSolver.hpp:134,223) re-evaluated independently with a power iteration on
          R = A V T V' + V T V' A' + B B' built from SpMM / Gram / panel products, and X = V T V' of the two back ends agreeing on
          random probes; configs[4]: SPD mass matrix (generalized residual by the same power iteration) and a warm start from the
          previous V after perturbing A's diagonal by 1 %.
Tolerances are fp64 rounding bounds and are written at each check."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M_ROWS = 1_000_000
PARAMS = {"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": 1e-6}


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=1)
    yield c
    c.close()


@pytest.fixture(scope="module")
def c3(ctx):
    import rails_amd
    from rails_amd import problems as P

    A = P.banded_random(M_ROWS, 27, 4096, seed=0)
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    B = P.rhs(M_ROWS, 16, seed=7)
    return A, op, B


def _fro2(X):
    return float(np.trace(X.dot(X)))


def test_spmm_full_size(ctx, oracle, c3):
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A, op, _ = c3
    rowptr, col, val = A
    m = M_ROWS
    # (1) the oracle on the whole matrix at the in-loop width
    g = np.random.default_rng(3)
    Xh = g.uniform(-1, 1, (m, 16))
    Y = op.apply(MV(ctx, data=Xh))
    ref = oracle.csr_spmm(rowptr, col, val, Xh)
    scale = np.abs(ref).max()
    assert np.abs(Y.to_host() - ref).max() <= 4e-14 * np.sqrt(27) * scale
    # (2) 128 columns (the headline SpMM): sampled rows against a host evaluation of the same rows
    X = MV(ctx, m=m, n=128)
    X.random()
    assert op.prepare(128)  # set-up for repeated products of this width (rails_csr_prepare): the sweep kernel's schedule
    Y = op.apply(X)
    assert op.last_kernel().startswith("k_spmm_sweep")  # banded pattern at panel width: the sweep kernel (spmm_sweep.hip) is the automatic choice
    Xh = X.to_host()
    rows = np.unique(np.concatenate([np.arange(0, 64), np.arange(m - 64, m), g.integers(0, m, 4000)]))
    Yh = Y.to_host()
    for i in rows[:: max(1, rows.size // 1500)]:
        p0, p1 = rowptr[i], rowptr[i + 1]
        want = val[p0:p1] @ Xh[col[p0:p1], :]
        assert np.abs(Yh[i] - want).max() <= 4e-14 * np.sqrt(27) * np.abs(val[p0:p1]).sum()
    del Yh
    # (3) linearity
    Z = MV(ctx, m=m, n=128)
    Z.random()
    W = 0.7 * X
    W -= 1.3 * Z
    YW = op.apply(W)
    comb = 0.7 * Y
    comb -= 1.3 * op.apply(Z)
    n2 = _fro2(YW)
    YW -= comb
    assert _fro2(YW) <= (1e-14) ** 2 * 27 * n2
    # (4) adjoint identity: Z' (A X) = (A' Z)' X
    G1 = Z.dot(Y)
    G2 = op.transpose().apply(Z).dot(X)
    assert np.abs(G1 - G2).max() <= 1e-12 * np.abs(G1).max()
    # (5) A * ones = row sums (the matrix is strictly diagonally dominant with row sum -1: rails_amd/problems.py)
    ones = MV(ctx, m=m, n=1)
    ones.assign(1.0)
    r = op.apply(ones).to_host()[:, 0]
    want = np.add.reduceat(val, rowptr[:-1])
    assert np.abs(r - want).max() <= 1e-13 * np.abs(val).max() * 27


def test_orthogonalize_and_projection_full_size(ctx):
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    m, k = M_ROWS, 200
    V = MV(ctx, m=m, n=k)
    V.random()
    V.orthogonalize()
    G = V.dot(V)
    assert np.abs(G - np.eye(k)).max() <= 1e-13  # block CGS2 + CholQR2: orthonormal to rounding
    # idempotence: orthogonalising an orthonormal panel changes nothing beyond rounding
    V0 = V.copy()
    V.orthogonalized = 0
    V.orthogonalize()
    V0 -= V
    assert _fro2(V0) <= (1e-13) ** 2 * k
    # X in span(V): V (V' X) = X
    g = np.random.default_rng(5)
    Cm = g.uniform(-1, 1, (k, 16))
    X = V.matmul(Cm)
    back = V.matmul(V.dot(X))
    n2 = _fro2(X)
    back -= X
    assert _fro2(back) <= (1e-13) ** 2 * n2


def _residual_norm_by_power_iteration(ctx, op, B, V, T, steps=30):
    """||R||_2 for R = A X + X A' + B B', X = V T V', from products with R only (symmetric: power iteration on R)"""
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    Vd, Bd = MV(ctx, data=V), MV(ctx, data=B)
    AV = op.apply(Vd)
    z = MV(ctx, m=V.shape[0], n=1)
    z.random()
    lam = 0.0
    for _ in range(steps):
        nz = np.sqrt(_fro2(z))
        z *= 1.0 / nz
        y = AV.matmul(T @ Vd.dot(z))       # A V T V' z
        y += Vd.matmul(T @ AV.dot(z))      # V T V' A' z
        y += Bd.matmul(Bd.dot(z))          # B B' z
        lam = np.sqrt(_fro2(y))
        z = y
    return lam


def test_solver_full_size_both_back_ends(ctx, c3):
    import rails_amd
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A, op, B = c3
    r0 = float(np.linalg.norm(B.T @ B, 2))  # ||B||_2^2
    out = {}
    for subspace in (1, 0):
        ctx.set_seed(1, 0)
        s = rails_amd.Solver(ctx, op, B)
        assert s.set_parameters(PARAMS) == 0
        s.set_option("verbose", 0)
        s.set_option("subspace", subspace)
        code, V, T = s.solve()
        assert code == 0
        k = V.shape[1]
        assert k <= PARAMS["Restart size"]
        assert np.abs(T - T.T).max() <= 1e-12 * np.abs(T).max()
        Vd = MV(ctx, data=V)
        # the reference does not re-orthogonalise after restarts (:270) either.  Measured (tests/diag_bounds.py, three seeds): 3e-15 .. 5e-15
        # on the coordinate-space back end; 0.07e-10 .. 1.2e-10 on the direct one, whose restarts rotate V in place with the hand-written
        # panel GEMM (the vendor GEMM is opt-in since round 3 and no test asks for it) -- the CholQR of the nearly dependent A*V blocks
        # sets that level, the rotation preserves it
        assert np.abs(Vd.dot(Vd) - np.eye(k)).max() <= (1e-12 if subspace else 2e-10)
        assert s.relative_residual() < PARAMS["Tolerance"]
        # the reference's own acceptance, re-evaluated independently: ||R||_2 < tol * ||B||_2^2.  The solver's value is a
        # 20-step Lanczos estimate (a lower bound that is tight for the dominant eigenvalue); allow it a factor 2.
        rn = _residual_norm_by_power_iteration(ctx, op, B, V, T)
        assert rn < 2.0 * PARAMS["Tolerance"] * r0
        out[subspace] = (V, T, s.trips())
        s.close()
        del Vd
    # both back ends solve the same equation to the same tolerance: X z agrees on random probes
    g = np.random.default_rng(9)
    Z = g.standard_normal((M_ROWS, 4))
    Xz = {b: V @ (T @ (V.T @ Z)) for b, (V, T, _) in out.items()}
    rel = np.linalg.norm(Xz[1] - Xz[0]) / np.linalg.norm(Xz[0])
    assert rel <= 20 * PARAMS["Tolerance"]
    assert abs(out[1][2] - out[0][2]) <= 3  # trips


def test_spmm_full_size_stencil_slab(ctx, oracle):
    """one GPU's share of configs[3]: 100^3 rows of the 27-point stencil (center -26, neighbours +1)"""
    import rails_amd
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A = P.stencil27(100, 100, 100)
    rowptr, col, val = A
    m = rowptr.size - 1
    assert m == M_ROWS
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    g = np.random.default_rng(4)
    Xh = g.uniform(-1, 1, (m, 32))  # configs[3]'s Expand size
    Y = op.apply(MV(ctx, data=Xh))
    ref = oracle.csr_spmm(rowptr, col, val, Xh)
    assert np.abs(Y.to_host() - ref).max() <= 4e-14 * np.sqrt(27) * np.abs(ref).max()
    X = MV(ctx, m=m, n=128)
    X.random()
    Y = op.apply(X)
    assert op.last_kernel() == "k_spmm_planes"  # complete 27-point stencil: the plane-sweep kernel (spmm_planes.hip)
    Xh, Yh = X.to_host(), Y.to_host()
    rows = np.unique(np.concatenate([np.arange(0, 32), np.arange(m - 32, m), g.integers(0, m, 1500)]))
    for i in rows:
        p0, p1 = rowptr[i], rowptr[i + 1]
        want = val[p0:p1] @ Xh[col[p0:p1], :]
        assert np.abs(Yh[i] - want).max() <= 4e-14 * np.sqrt(27) * np.abs(val[p0:p1]).sum()
    # the operator is symmetric: Z' (A X) = (A Z)' X
    Z = MV(ctx, m=m, n=128)
    Z.random()
    G1, G2 = Z.dot(Y), op.apply(Z).dot(X)
    assert np.abs(G1 - G2).max() <= 1e-12 * np.abs(G1).max()


def test_solver_full_size_mass_matrix_and_warm_start(ctx, c3):
    """configs[4]: m = 1M, SPD M, warm start from the previous solve"""
    import rails_amd
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A, op, B = c3
    Mcsr = P.mass_diag(M_ROWS, seed=11)
    Mop = rails_amd.HipOperatorWrapper(ctx, *Mcsr)
    params = dict(PARAMS, Tolerance=1e-4)
    r0 = float(np.linalg.norm(B.T @ B, 2))

    def solve(A_op, V0=None, extra=None):
        ctx.set_seed(1, 0)
        s = rails_amd.Solver(ctx, A_op, B, M=Mop)
        assert s.set_parameters(dict(params, **(extra or {}))) == 0
        s.set_option("verbose", 0)
        s.set_option("mass", 1)
        code, V, T = s.solve(V0=V0)
        assert code == 0
        trips = s.trips()
        s.close()
        return V, T, trips

    def generalized_residual_norm(A_op, V, T, steps=30):
        Vd, Bd = MV(ctx, data=V), MV(ctx, data=B)
        AV, MV_ = A_op.apply(Vd), Mop.apply(Vd)
        z = MV(ctx, m=M_ROWS, n=1)
        z.random()
        lam = 0.0
        for _ in range(steps):
            z *= 1.0 / np.sqrt(_fro2(z))
            y = AV.matmul(T @ MV_.dot(z))   # A V T V' M' z
            y += MV_.matmul(T @ AV.dot(z))  # M V T V' A' z
            y += Bd.matmul(Bd.dot(z))
            lam = np.sqrt(_fro2(y))
            z = y
        return lam

    V, T, cold = solve(op)
    assert generalized_residual_norm(op, V, T) < 2.0 * params["Tolerance"] * r0
    # perturb A's diagonal by 1 %, continue from V ("Restart from solution", src/LyapunovSolver.hpp:116-123)
    rowptr, col, val = A
    val2 = val.copy()
    val2[col == np.repeat(np.arange(M_ROWS), np.diff(rowptr))] *= 1.01
    op2 = rails_amd.HipOperatorWrapper(ctx, rowptr, col, val2)
    V2, T2, warm = solve(op2, V0=V, extra={"Restart from solution": 1})
    assert warm < cold
    assert generalized_residual_norm(op2, V2, T2) < 2.0 * params["Tolerance"] * r0
