"""Schur-complement operator for a singular diagonal mass matrix (rails_amd/schur.py; reference: src/SchurOperator.cpp:51-214) and
the operator-callback handle of the C ABI it plugs in through (rails_csr_create_callback)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descriptor_system(n1=40, n2=150, p=3, seed=0):
    import scipy.sparse as sp

    g = np.random.default_rng(seed)
    n = n1 + n2
    perm = g.permutation(n)
    set1 = np.sort(perm[:n1])  # constraint unknowns scattered through the index range
    mask1 = np.zeros(n, dtype=bool)
    mask1[set1] = True
    A = sp.random(n, n, density=0.04, random_state=np.random.RandomState(seed), format="lil")
    A = (0.3 * A).tolil()
    A.setdiag(np.where(mask1, 2.0 + g.uniform(0, 1, n), -4.0 - g.uniform(0, 1, n)))
    A = A.tocsr()
    A.sort_indices()
    mass = np.where(mask1, 0.0, 1.0)
    B = g.uniform(-1, 1, (n, p))
    B[mask1] = 0.0
    return (A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)), A.toarray(), mass, B, mask1


def test_operator_callback_handle():
    import rails_amd
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    ctx = rails_amd.Context(device=0, seed=2)
    m = 500
    g = np.random.default_rng(1)
    D = g.uniform(1, 2, m)
    calls = []

    def apply(trans, X, Y):  # Y = diag(D) X through host memory: the callback sees windows of the caller's panels
        calls.append((trans, X.n, X.c0, Y.c0))
        Y.from_host(np.asfortranarray(D[:, None] * X.to_host()))
        return 0

    op = rails_amd.HipOperatorWrapper.from_callback(ctx, m, apply)
    assert op.M() == m and op.nnz() == 0 and op.last_kernel() == "callback"
    Xh = g.uniform(-1, 1, (m, 7))
    big = MV(ctx, m=m, n=10, capacity=12)
    X = big.view(2, 8)
    X.from_host(Xh)
    out = MV(ctx, m=m, n=9, capacity=9)
    Y = out.view(1, 7)
    op.apply(X, Y)
    np.testing.assert_allclose(Y.to_host(), D[:, None] * Xh, rtol=0, atol=0)
    assert calls[-1] == (False, 7, 2, 1)
    op.transpose().apply(X, Y)
    assert calls[-1][0] is True
    assert ctx.stats()["spmm_callback"] == 2

    def failing(trans, X, Y):
        raise RuntimeError("boom")

    bad = rails_amd.HipOperatorWrapper.from_callback(ctx, m, failing)
    with pytest.raises(rails_amd.RailsError):
        bad.apply(X, Y)
    ctx.close()


def test_rectangular_operators():
    """rails_csr_create_rect: wide and tall CSR operators (the blocks A12, A21 of the Schur complement) against scipy, at the in-loop widths"""
    import scipy.sparse as sp

    import rails_amd
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    ctx = rails_amd.Context(device=0, seed=1)
    g = np.random.default_rng(11)
    for n_rows, n_cols in ((700, 2300), (2300, 700), (1, 50), (50, 1)):
        Mx = sp.random(n_rows, n_cols, density=min(1.0, 12.0 / n_cols), random_state=3, format="csr")
        Mx.sort_indices()
        op = rails_amd.HipOperatorWrapper.rect(ctx, n_rows, n_cols, Mx.indptr.astype(np.int64), Mx.indices.astype(np.int32), Mx.data)
        for nc in (1, 5, 16, 40):
            Xh = g.uniform(-1, 1, (n_cols, nc))
            Y = op.apply(MV(ctx, data=Xh))
            assert Y.to_host().shape == (n_rows, nc)
            ref = Mx @ Xh
            assert np.abs(Y.to_host() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
        with pytest.raises(rails_amd.RailsError):
            op.apply(MV(ctx, data=g.uniform(-1, 1, (n_cols + 1, 2))))  # X has to have n_cols rows
        with pytest.raises(rails_amd.RailsError):
            op.transpose().apply(MV(ctx, data=g.uniform(-1, 1, (n_cols, 2))))  # no transposed apply: create the transposed matrix
    ctx.close()


def test_sparse_triangular_solves_and_device_lu():
    """rails_sptrsv_solve / DeviceLU (rails_amd/csrc/sptrsv.hip; what src/SchurOperator.cpp:193-200 does on the host with the KLU factors):
    the L and U factors of scipy's SuperLU applied on the device against the host solve -- a random sparse matrix (few, wide levels), a
    banded one (hundreds of narrow levels: the one-workgroup chain), both transposes, panel windows, widths 1..40."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    import rails_amd
    from rails_amd.schur import DeviceLU
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    ctx = rails_amd.Context(device=0, seed=6)
    g = np.random.default_rng(4)
    cases = []
    n = 1500
    cases.append(("random", sp.random(n, n, density=4.0 / n, random_state=7, format="csc") + sp.identity(n, format="csc") * 3.0))
    n = 900
    band = sp.diags([g.uniform(-1, 1, n - abs(k)) for k in (-7, -2, -1, 0, 1, 3, 9)], (-7, -2, -1, 0, 1, 3, 9), format="lil")
    band.setdiag(4.0 + g.uniform(0, 1, n))
    cases.append(("banded", band.tocsc()))
    cases.append(("tiny", sp.csc_matrix(np.array([[2.0]]))))
    for name, A in cases:
        n = A.shape[0]
        lu = spla.splu(A.tocsc())
        dlu = DeviceLU(ctx, lu)
        lv = dlu.levels()
        print(name, n, "levels", lv)
        if name == "banded":
            assert lv["L"] > 100 and lv["U"] > 100
        for nc, off in ((1, 0), (5, 2), (16, 0), (40, 3)):
            Bh = g.uniform(-1, 1, (n, nc))
            big = MV(ctx, m=n, n=nc + off, capacity=nc + off + 2)
            B = big.view(off, off + nc - 1)
            B.from_host(Bh)
            tmp, out = MV(ctx, m=n, n=nc, capacity=nc), MV(ctx, m=n, n=nc + 1, capacity=nc + 1)
            for trans in (False, True):
                dlu.solve(B, tmp, out.view(1, nc), trans=trans)
                ref = lu.solve(Bh, trans="T" if trans else "N").reshape(n, nc)
                got = out.view(1, nc).to_host()
                assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (name, nc, trans, np.abs(got - ref).max())
                assert np.array_equal(B.to_host(), Bh)  # the right-hand side is left alone
        dlu.close()
    # what the library refuses: an entry outside the triangle, a missing or zero diagonal, a stored diagonal of a unit triangle
    import ctypes as C

    def create(rp, ci, va, lower, unit):
        h = C.c_void_p()
        rp, ci, va = np.asarray(rp, np.int64), np.asarray(ci, np.int32), np.asarray(va, np.float64)
        return ctx.lib.rails_sptrsv_create(ctx.h, rp.size - 1, rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                           va.ctypes.data_as(C.POINTER(C.c_double)), lower, unit, C.byref(h))

    assert create([0, 2, 3], [0, 1, 1], [1.0, 2.0, 1.0], 1, 0) != 0  # (0, 1) in a lower triangle
    assert create([0, 1, 2], [0, 0], [1.0, 2.0], 1, 0) != 0  # row 1 without a diagonal
    assert create([0, 1, 2], [0, 1], [1.0, 0.0], 1, 0) != 0  # zero pivot
    assert create([0, 1, 2], [0, 1], [1.0, 1.0], 1, 1) != 0  # unit triangle with its diagonal stored
    ctx.close()


@pytest.mark.parametrize("subspace,device_solve", [(1, True), (0, True), (1, False)])
def test_schur_operator_and_solve(subspace, device_solve):
    import scipy.linalg as sl

    import rails_amd
    from rails_amd.schur import SchurOperator
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A, Ad, mass, B, mask1 = _descriptor_system()
    ctx = rails_amd.Context(device=0, seed=4)
    S = SchurOperator(ctx, A, mass, device_solve=device_solve)
    i1, i2 = np.flatnonzero(mask1), np.flatnonzero(~mask1)
    assert np.array_equal(S.idx1, i1) and np.array_equal(S.idx2, i2)
    Sd = Ad[np.ix_(i2, i2)] - Ad[np.ix_(i2, i1)] @ np.linalg.solve(Ad[np.ix_(i1, i1)], Ad[np.ix_(i1, i2)])
    np.testing.assert_allclose(S.dense(), Sd, atol=1e-12)
    g = np.random.default_rng(2)
    Xh = g.uniform(-1, 1, (S.m2, 5))
    Y = S.op.apply(MV(ctx, data=Xh))
    np.testing.assert_allclose(Y.to_host(), Sd @ Xh, atol=1e-12)
    Yt = S.op.transpose().apply(MV(ctx, data=Xh))
    np.testing.assert_allclose(Yt.to_host(), Sd.T @ Xh, atol=1e-12)
    assert S.applies == 10  # matrix-vector products through the operator, as SchurOperator::GetMVPs counts them
    # X stayed on the device.  With the solve on the device nothing of a product crossed PCIe; with the host solve only the m1 x nc block
    # A12 X went to the host and the solution of the A11 system came back
    assert S.host_bytes == (0 if device_solve else 2 * 2 * S.m1 * 5 * 8) and S.m1 < S.m2
    # the Lyapunov equation on the Schur complement (what src/main.cpp:90-118 sets up): S X + X S' + B2 B2' = 0
    B2 = S.restrict(B)
    s = rails_amd.Solver(ctx, S.op, B2)
    assert s.set_parameters({"Restart size": 60, "Reduced size": 30, "Expand size": 3, "Lanczos iterations": 5, "Tolerance": 1e-8}) == 0
    s.set_option("verbose", 0)
    s.set_option("subspace", subspace)
    code, V, T = s.solve()
    assert code == 0
    X = V @ T @ V.T
    Xref = sl.solve_continuous_lyapunov(Sd, -B2 @ B2.T)
    assert np.linalg.norm(X - Xref) / np.linalg.norm(Xref) < 1e-6
    assert s.relative_residual() < 1e-6
    assert bool(s.backend_stats()) == bool(subspace)
    # a general nonsingular part of the mass matrix: S X D + D X S' + B2 B2' = 0 with D = diag(M22)
    mass2 = mass.copy()
    mass2[~mask1] = g.uniform(0.5, 1.5, (~mask1).sum())
    S2 = SchurOperator(ctx, A, mass2, device_solve=device_solve)
    Dm = S2.mass22
    Mop = rails_amd.HipOperatorWrapper(ctx, np.arange(S2.m2 + 1, dtype=np.int64), np.arange(S2.m2, dtype=np.int32), Dm)
    s2 = rails_amd.Solver(ctx, S2.op, B2, M=Mop)
    assert s2.set_parameters({"Restart size": 60, "Reduced size": 30, "Expand size": 3, "Lanczos iterations": 5, "Tolerance": 1e-8}) == 0
    s2.set_option("verbose", 0)
    s2.set_option("mass", 1)
    s2.set_option("subspace", subspace)
    code, V2, T2 = s2.solve()
    assert code == 0
    X2 = V2 @ T2 @ V2.T
    R = Sd @ X2 * Dm[None, :] + (Dm[:, None] * X2) @ Sd.T + B2 @ B2.T
    assert np.linalg.norm(R) / np.linalg.norm(B2 @ B2.T) < 1e-6
    s.close()
    s2.close()
    ctx.close()


def test_driver_with_singular_mass_matrix(tmp_path):
    """python -m rails_amd.main with an M.mtx that has zero diagonal entries: the Schur-complement route of src/main.cpp:77-118"""
    import os
    import subprocess
    import sys

    import scipy.linalg as sl

    from rails_amd import mmio

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A, Ad, mass, B, mask1 = _descriptor_system(n1=30, n2=120, p=2, seed=3)
    n = mass.size
    mmio.write_csr(str(tmp_path / "A.mtx"), n, n, *A)
    mmio.write_csr(str(tmp_path / "M.mtx"), n, n, np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), mass)  # explicit zeros
    mmio.write_array(str(tmp_path / "B.mtx"), B)
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "rails_amd.main", "--dir", str(tmp_path), "--set", "Tolerance=1e-8", "--set", "Restart size=60",
                        "--set", "Reduced size=30", "--set", "Expand size=3", "--set", "Lanczos iterations=5"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "Computing Schur complement" in p.stdout and "matrix-vector products" in p.stdout
    V = mmio.read_dense(str(tmp_path / "V.mtx"))
    T = mmio.read_dense(str(tmp_path / "T.mtx"))
    i1, i2 = np.flatnonzero(mask1), np.flatnonzero(~mask1)
    assert V.shape[0] == i2.size
    Sd = Ad[np.ix_(i2, i2)] - Ad[np.ix_(i2, i1)] @ np.linalg.solve(Ad[np.ix_(i1, i1)], Ad[np.ix_(i1, i2)])
    Xref = sl.solve_continuous_lyapunov(Sd, -B[i2] @ B[i2].T)
    X = V @ T @ V.T
    assert np.linalg.norm(X - Xref) / np.linalg.norm(Xref) < 1e-6
