"""Parity of the HIP kernels (through the C ABI) against the CPU oracle on identical inputs.

fp64 tolerances (stated per test): integer/index work is exact; SpMM <= 1e-14*sqrt(nnz/row) relative;
row reductions (Gram, Lanczos sums) differ from the oracle only by summation order:
<= 1e-13 * sqrt(m)-scaled bounds.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=1234)
    yield c
    c.close()


def MV(ctx, data=None, **kw):
    import rails_amd

    return rails_amd.HipMultiVectorWrapper(ctx, data=data, **kw)


def test_transfer_and_blas1(ctx):
    g = np.random.default_rng(0)
    for m, n in ((1, 1), (5, 3), (257, 17), (4099, 40)):
        X = g.uniform(-1, 1, (m, n))
        a = MV(ctx, X, capacity=n + 7)
        assert np.array_equal(a.to_host(), X)
        b = a.copy()
        b *= 2.5
        assert np.array_equal(b.to_host(), X * 2.5)
        b += a
        np.testing.assert_array_equal(b.to_host(), X * 2.5 + X)
        b -= a
        b /= 13.0
        np.testing.assert_array_equal(b.to_host(), (X * 2.5 + X - X) * (1.0 / 13.0))
        # views write through, also at odd column offsets
        if n >= 3:
            v = a.view(1, 2)
            v.assign(7.0)
            Xe = X.copy()
            Xe[:, 1:3] = 7.0
            assert np.array_equal(a.to_host(), Xe)
        a.resize(n + 5)  # within capacity: data preserved (src/StlWrapper.cpp:231-236)
        assert np.array_equal(a.to_host()[:, 0], (Xe if n >= 3 else X)[:, 0])
        a.resize(n + 40)  # beyond capacity: re-allocation preserves data (:238-248)
        assert np.array_equal(a.to_host()[:, 0], (Xe if n >= 3 else X)[:, 0])


def test_random_matches_counter_generator(ctx, oracle):
    ctx.set_seed(99, 5)
    a = MV(ctx, m=1000, n=6, capacity=8)
    a.random()
    assert np.array_equal(a.to_host(), oracle.random(1000, 6, mode=1, seed=99, stream=5))
    v = a.view(2)
    v.random()  # next stream id, column index restarts at 0 inside the view
    assert np.array_equal(v.to_host(), oracle.random(1000, 1, mode=1, seed=99, stream=6))
    assert np.abs(a.to_host()).max() < 1.0


def _csr_cases():
    from rails_amd import problems as P

    return {
        "laplace7_small": P.laplace7(7, 5, 4),
        "stencil27_rand": P.stencil27(9, 8, 7, random_values=True, seed=3),
        "banded": P.banded_random(5000, 27, 300, seed=1),
        "uniform": P.uniform_random(3001, 11, seed=2),
        "dense": P.dense_to_csr(np.random.default_rng(5).uniform(-1, 1, (70, 70))),
    }


@pytest.mark.parametrize("name", ["laplace7_small", "stencil27_rand", "banded", "uniform", "dense"])
def test_spmm_matches_oracle(ctx, oracle, name):
    import rails_amd

    A = _csr_cases()[name]
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    g = np.random.default_rng(11)
    nnz_row = max(1, A[1].size // m)
    for nc, xoff, yoff in ((1, 0, 0), (3, 0, 0), (8, 0, 2), (16, 16, 0), (16, 0, 3), (11, 2, 5), (17, 1, 0), (64, 0, 1), (128, 0, 0), (130, 2, 4)):
        Xh = g.uniform(-1, 1, (m, nc))
        big = MV(ctx, m=m, n=nc + xoff, capacity=nc + xoff)
        X = big.view(xoff, xoff + nc - 1) if nc > 1 else big.view(xoff)
        X.from_host(Xh)
        outp = MV(ctx, m=m, n=nc + yoff, capacity=nc + yoff + 3)
        Y = outp.view(yoff, yoff + nc - 1) if nc > 1 else outp.view(yoff)
        op.apply(X, Y)
        ref = oracle.csr_spmm(*A, Xh)
        scale = np.abs(ref).max() + 1e-300
        assert np.abs(Y.to_host() - ref).max() <= 1e-14 * np.sqrt(nnz_row) * scale * 4
        # transposed apply (GenericOperatorWrapper_test.cpp:91-109)
        opT = op.transpose()
        YT = opT.apply(X)
        rowptr, col, val = A
        dense = np.zeros((m, m)) if m <= 400 else None
        if dense is not None:
            for i in range(m):
                for p in range(rowptr[i], rowptr[i + 1]):
                    dense[i, col[p]] += val[p]
            np.testing.assert_allclose(YT.to_host(), dense.T @ Xh, atol=1e-13 * scale * nnz_row)


@pytest.mark.parametrize("variant", [2, 6])
@pytest.mark.parametrize("name", ["laplace7_small", "stencil27_rand", "banded"])
def test_spmm_lds_staged_kernel_matches_oracle(ctx, oracle, name, variant):
    """variant 2 = LDS-staged footprint kernel (k_spmm_tiled*), 6 = the same with 16-column chunks; the same sums in the same
    order per row as the row-gather kernel."""
    import rails_amd
    from rails_amd import problems as P

    A = {"laplace7_small": P.laplace7(23, 11, 9), "stencil27_rand": P.stencil27(20, 12, 9, random_values=True, seed=3),
         "banded": P.banded_random(5000, 27, 40, seed=1)}[name]
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    op.set_variant(variant)
    g = np.random.default_rng(13)
    for nc, xoff, yoff in ((8, 0, 0), (16, 16, 0), (17, 0, 2), (64, 2, 0), (128, 0, 0), (130, 0, 0)):
        Xh = g.uniform(-1, 1, (m, nc))
        big = MV(ctx, m=m, n=nc + xoff, capacity=nc + xoff)
        X = big.view(xoff, xoff + nc - 1)
        X.from_host(Xh)
        outp = MV(ctx, m=m, n=nc + yoff, capacity=nc + yoff + 3)
        Y = outp.view(yoff, yoff + nc - 1)
        op.apply(X, Y)
        assert op.last_kernel().startswith("k_spmm_tiled")
        ref = oracle.csr_spmm(*A, Xh)
        assert np.abs(Y.to_host() - ref).max() <= 1e-14 * np.abs(ref).max() * 8
    # no reuse between rows (uniform random columns): the kernel must refuse rather than run slowly
    opu = rails_amd.HipOperatorWrapper(ctx, *P.uniform_random(3001, 11, seed=2))
    opu.set_variant(2)
    with pytest.raises(rails_amd.RailsError):
        opu.apply(MV(ctx, g.uniform(-1, 1, (3001, 16))))


@pytest.mark.parametrize("variant", [4, 5])
def test_spmm_rowgather_column_chunks_match_oracle(ctx, oracle, variant):
    """variants 4 / 5 = row-gather in 32 / 64 column chunks inside one launch (k_spmm_rowgather_cc): same per-row summation
    order as the plain row-gather kernel; ragged rows, a row block longer than the LDS staging buffer, odd widths."""
    import rails_amd
    from rails_amd import problems as P

    g = np.random.default_rng(17)
    # ragged: empty rows, rows of 1..300 entries (blocks of 64 rows exceed the 2048-entry LDS buffer -> global (col, val) path)
    m = 777
    lens = g.integers(0, 12, m)
    lens[100:140] = g.integers(100, 300, 40)
    lens[5] = 0
    lens[-1] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = g.integers(0, m, rowptr[-1]).astype(np.int32)
    val = g.uniform(-1, 1, rowptr[-1])
    cases = [P.banded_random(5000, 27, 300, seed=1), P.stencil27(9, 8, 7, random_values=True, seed=3), (rowptr, col, val)]
    for A in cases:
        mm = A[0].size - 1
        op = rails_amd.HipOperatorWrapper(ctx, *A)
        op.set_variant(variant)
        for nc, xoff, yoff in ((128, 0, 0), (130, 2, 4), (97, 0, 0), (66, 0, 2), (40, 0, 0)):
            if nc <= 32 * (variant - 3):
                continue
            Xh = g.uniform(-1, 1, (mm, nc))
            big = MV(ctx, m=mm, n=nc + xoff, capacity=nc + xoff)
            X = big.view(xoff, xoff + nc - 1)
            X.from_host(Xh)
            outp = MV(ctx, m=mm, n=nc + yoff, capacity=nc + yoff + 3)
            Y = outp.view(yoff, yoff + nc - 1)
            op.apply(X, Y)
            assert op.last_kernel() == "k_spmm_rowgather_cc"
            ref = oracle.csr_spmm(*A, Xh)
            assert np.abs(Y.to_host() - ref).max() <= 1e-14 * np.abs(ref).max() * 40
            op.set_variant(3)
            Y3 = op.apply(X)
            assert op.last_kernel() == "k_spmm_rowgather"
            assert np.array_equal(Y3.to_host(), Y.to_host())  # identical summation order -> bit-identical
            op.set_variant(variant)


def test_spmm_ragged_and_empty_rows(ctx, oracle):
    import rails_amd

    # rows with 0, 1 and many entries; last row empty
    rowptr = np.array([0, 0, 1, 1, 40, 43, 43], dtype=np.int64)
    g = np.random.default_rng(2)
    col = g.integers(0, 6, 43).astype(np.int32)
    val = g.uniform(-1, 1, 43)
    op = rails_amd.HipOperatorWrapper(ctx, rowptr, col, val)
    Xh = g.uniform(-1, 1, (6, 5))
    Y = op.apply(MV(ctx, Xh))
    np.testing.assert_allclose(Y.to_host(), oracle.csr_spmm(rowptr, col, val, Xh), atol=1e-14)


def test_csr_create_rejects_bad_indices(ctx):
    import rails_amd

    with pytest.raises(rails_amd.RailsError):
        rails_amd.HipOperatorWrapper(ctx, np.array([0, 1]), np.array([5], dtype=np.int32), np.array([1.0]))


@pytest.mark.parametrize("m", [1, 63, 1000, 20011])
def test_gram_matches_oracle(ctx, oracle, m):
    g = np.random.default_rng(m)
    # (130 x 17, 367 x 17: the first Gram matrix of an A*W block with its 17th column, k_gram_cols<TI, 2>)
    for a, b in ((1, 1), (3, 5), (16, 16), (40, 8), (130, 16), (16, 130), (64, 33), (17, 1), (130, 17), (367, 17), (200, 18), (150, 20)):
        Xh = g.uniform(-1, 1, (m, a))
        Yh = g.uniform(-1, 1, (m, b))
        X, Y = MV(ctx, Xh), MV(ctx, Yh)
        C = X.dot(Y)
        ref = oracle.dot(Xh, Yh)
        assert C.shape == (a, b)
        assert np.abs(C - ref).max() <= 1e-14 * m + 1e-13 * np.sqrt(m)


def test_dot_on_views_and_norm(ctx, oracle):
    g = np.random.default_rng(4)
    Xh = g.uniform(-1, 1, (3000, 9))
    X = MV(ctx, Xh)
    C = X.view(1, 3).dot(X.view(4, 8))
    np.testing.assert_allclose(C, Xh[:, 1:4].T @ Xh[:, 4:9], atol=1e-11)
    # norm() = spectral 2-norm (F7), single column = Euclidean
    assert abs(X.norm() - oracle.norm2(Xh)) < 1e-11
    assert abs(X.view(2).norm() - np.linalg.norm(Xh[:, 2])) < 1e-11


@pytest.mark.parametrize("m", [1, 50, 4097])
def test_panel_gemm_matches_oracle(ctx, oracle, m):
    g = np.random.default_rng(m + 1)
    for k, r in ((1, 1), (5, 3), (16, 16), (33, 7), (64, 64), (200, 128), (20, 16), (129, 40), (40, 200)):
        Xh = g.uniform(-1, 1, (m, k))
        Ch = g.uniform(-1, 1, (k, r))
        X = MV(ctx, Xh)
        Y = X.matmul(Ch)
        ref = oracle.panel_gemm(Xh, Ch)
        assert np.abs(Y.to_host() - ref).max() <= 1e-14 * k * 4
        # beta != 0 and alpha != 1 into an offset window
        Y0 = g.uniform(-1, 1, (m, r + 1))
        out = MV(ctx, Y0, capacity=r + 1)
        X.gemm_into(Ch, out.view(1, r) if r > 1 else out.view(1), alpha=-1.0, beta=1.0)
        ref2 = Y0.copy()
        ref2[:, 1:] -= Xh @ Ch
        assert np.abs(out.to_host() - ref2).max() <= 1e-14 * k * 4 + 1e-15
    # in place: V.view(0, r-1) = V * X  (src/LyapunovSolver.hpp:265)
    Xh = g.uniform(-1, 1, (m, 40))
    Ch = g.uniform(-1, 1, (40, 24))
    X = MV(ctx, Xh)
    X.gemm_into(Ch, X.view(0, 23))
    np.testing.assert_allclose(X.to_host()[:, :24], Xh @ Ch, atol=1e-12)
    np.testing.assert_array_equal(X.to_host()[:, 24:], Xh[:, 24:])


def test_orthogonalize_kat(ctx):
    # test/GenericMultiVectorWrapper_test.cpp:270-311
    a = np.zeros((10, 2))
    a[0, 0], a[0, 1], a[1, 1] = 2.3, 5.3, 2.7
    e = np.zeros((10, 2))
    e[0, 0] = e[1, 1] = 1.0
    for method in (0, 1, 2):
        X = MV(ctx, a)
        X.orthogonalize(method)
        np.testing.assert_allclose(X.to_host(), e, atol=4e-16)
    # watermark: orthogonalize one column, push_back, orthogonalize again
    X = MV(ctx, a[:, :1], capacity=4)
    X.orthogonalize()
    c = np.zeros((10, 1))
    c[0, 0], c[1, 0] = 5.3, 2.7
    X.push_back(MV(ctx, c))
    X.orthogonalize()
    np.testing.assert_allclose(X.to_host(), e, atol=4e-16)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_orthogonalize_matches_oracle(ctx, oracle, method):
    g = np.random.default_rng(8)
    m = 6000
    V1 = g.uniform(-1, 1, (m, 20))
    V2 = g.uniform(-1, 1, (m, 16))
    X = MV(ctx, V1, capacity=40)
    used = X.orthogonalize(method)
    assert used == (1 if method == 1 else 2)
    o1 = oracle.orthogonalize(V1)
    np.testing.assert_allclose(X.to_host(), o1, atol=1e-12)
    X.push_back(MV(ctx, V2))
    X.orthogonalize(method)
    Q = X.to_host()
    o12 = oracle.orthogonalize(np.hstack([o1, V2]), start=20)
    np.testing.assert_allclose(Q, o12, atol=1e-11)
    assert np.abs(Q.T @ Q - np.eye(36)).max() < 5e-15 * np.sqrt(m)


def test_orthogonalize_rank_deficient_falls_back(ctx):
    g = np.random.default_rng(9)
    m = 2000
    V1 = np.linalg.qr(g.uniform(-1, 1, (m, 6)))[0]
    # (a) a new column that lies in span(V_old) up to 1e-13: after projection it is pure rounding noise, which
    # both the reference's recurrence and the block form normalise (src/StlWrapper.cpp:318); the other columns
    # must come out orthonormal either way
    W = np.hstack([g.uniform(-1, 1, (m, 2)), V1[:, :1] + 1e-13 * g.uniform(-1, 1, (m, 1))])
    X = MV(ctx, np.hstack([V1, W]), capacity=12)
    X.orthogonalized = 6
    used = X.orthogonalize(0)
    assert used in (1, 2, 3)
    Q = X.to_host()
    assert np.abs(Q[:, :8].T @ Q[:, :8] - np.eye(8)).max() < 1e-13
    assert np.all(np.isfinite(Q))
    # (b) two identical new columns: the block Gram matrix is singular -> column-wise recurrence of the reference
    w = g.uniform(-1, 1, (m, 1))
    X = MV(ctx, np.hstack([V1, w, w, g.uniform(-1, 1, (m, 1))]), capacity=12)
    X.orthogonalized = 6
    used = X.orthogonalize(0)
    assert used == 3  # block method after one repair round (orth.hip: repair_block)
    Q = X.to_host()
    assert np.all(np.isfinite(Q))
    keep = [0, 1, 2, 3, 4, 5, 6, 8]
    assert np.abs(Q[:, keep].T @ Q[:, keep] - np.eye(8)).max() < 1e-13
    assert np.abs(Q.T @ Q - np.eye(9)).max() < 1e-13  # the dependent column became a unit vector orthogonal to the rest
    # (c) the RAILS pattern: expansion vectors in +/- pairs that coincide after span(V) is projected out (8 of 16 dependent);
    # independent columns must equal the reference's recurrence, the whole block must come out orthonormal
    Z = g.uniform(-1, 1, (m, 8))
    C1, C2 = g.uniform(-1, 1, (6, 8)), g.uniform(-1, 1, (6, 8))
    Wp = np.empty((m, 16))
    Wp[:, 0::2] = V1 @ C1 + Z
    Wp[:, 1::2] = V1 @ C2 - Z
    X = MV(ctx, np.hstack([V1, Wp]), capacity=24)
    X.orthogonalized = 6
    used = X.orthogonalize(0)
    assert used == 3
    Q = X.to_host()
    assert np.abs(Q.T @ Q - np.eye(22)).max() < 1e-12
    # the block spans the independent directions (as in the reference, each later column is also orthogonalised against the
    # normalised noise the dependent columns turned into, so the vectors themselves are not the QR factor of Z)
    assert np.abs(Z - Q @ (Q.T @ Z)).max() < 1e-10
    # RAILS_ORTH_REPAIR is read once per process; the column-wise fallback itself stays covered by method 1 above


def _lanczos_case(g, m, k, p):
    A = g.uniform(-1, 1, (m, 6))  # low-rank-ish operator pieces keep this cheap on the CPU
    V = np.linalg.qr(g.uniform(-1, 1, (m, k)))[0]
    AV = A @ (A.T @ V) / m - 2.0 * V + 0.1 * g.uniform(-1, 1, (m, k))
    B = g.uniform(-1, 1, (m, p))
    T = g.uniform(-1, 1, (k, k))
    T = 0.05 * (T + T.T)
    return AV, V, T, B


@pytest.mark.parametrize("m,k,p,L", [(240, 12, 4, 10), (5000, 40, 8, 12), (3001, 130, 16, 10), (777, 3, 1, 6)])
def test_resid_lanczos_matches_oracle(ctx, oracle, m, k, p, L):
    import rails_amd

    g = np.random.default_rng(m)
    AVh, Vh, T, Bh = _lanczos_case(g, m, k, p)
    AV, V, B = MV(ctx, AVh, capacity=k + 6), MV(ctx, Vh, capacity=k + 2), MV(ctx, Bh)
    ctx.set_seed(4242, 17)
    out = rails_amd.resid_lanczos(ctx, AV, V, T, B, L)
    ref = oracle.resid_lanczos(AVh, Vh, T, Bh, L, rng_mode=1, seed=4242, stream=17)
    n = ref["steps"]
    assert out["steps"] == n
    scale = np.abs(ref["H"]).max()
    # tridiagonal entries: same recurrence, different summation order -> compare to 1e-9 relative
    np.testing.assert_allclose(out["H"][:n, :n], ref["H"][:n, :n], rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(out["eigenvalues"], ref["eigenvalues"], rtol=0, atol=1e-9 * scale)
    # eigenvectors = Q * v  (src/LyapunovSolver.hpp:443), compared up to sign
    E = MV(ctx, m=m, n=n, capacity=n)
    rails_amd.lanczos_vectors(ctx, out["v"], E)
    Eh = E.to_host()
    R = ref["eigenvectors"]
    s = np.sign((Eh * R).sum(0))
    s[s == 0] = 1
    # Ritz vectors of well separated Ritz values only (clusters rotate freely)
    w = ref["eigenvalues"]
    gap = np.array([min(abs(w[i] - w[j]) for j in range(n) if j != i) for i in range(n)])
    good = gap > 1e-3 * scale
    assert good.sum() >= 1
    small_rank = 2 * k + p <= L + 2  # Krylov space (nearly) exhausted: Ritz vectors are ill-conditioned
    assert np.abs(Eh * s - R)[:, good].max() < (1e-4 if small_rank else 1e-7)
    # the Lanczos basis itself is orthonormal to the level the recurrence allows
    Q = MV(ctx, m=m, n=n, capacity=n)
    rails_amd.lanczos_vectors(ctx, np.eye(n), Q)
    nq = n - 2 if small_rank else n  # the last vectors of an exhausted Krylov space are rounding-level chaotic
    np.testing.assert_allclose(Q.to_host()[:, :nq], ref["Q"][:, :nq], atol=1e-6 if small_rank else 1e-8)


def test_resid_lanczos_breakdown(ctx, oracle):
    """beta < 1e-14 exit (src/LyapunovSolver.hpp:419-426): R of rank 1 from B alone (T = 0)."""
    import rails_amd

    m, k = 500, 2
    g = np.random.default_rng(1)
    Vh = np.linalg.qr(g.uniform(-1, 1, (m, k)))[0]
    AVh = g.uniform(-1, 1, (m, k))
    Bh = g.uniform(-1, 1, (m, 1))
    T = np.zeros((k, k))
    ctx.set_seed(7, 0)
    out = rails_amd.resid_lanczos(ctx, MV(ctx, AVh), MV(ctx, Vh), T, MV(ctx, Bh), 8)
    ref = oracle.resid_lanczos(AVh, Vh, T, Bh, 8, rng_mode=1, seed=7, stream=0)
    # rank-1 operator: the Krylov space is exhausted after 2 vectors; both implementations either break
    # down or continue on rounding noise.  The dominant Ritz value is ||B||^2 either way.
    assert abs(np.abs(out["eigenvalues"]).max() - float(Bh.T @ Bh)) < 1e-9 * float(Bh.T @ Bh)
    assert abs(np.abs(ref["eigenvalues"]).max() - float(Bh.T @ Bh)) < 1e-9 * float(Bh.T @ Bh)
    assert out["steps"] <= 8


def test_panel_gemm_wide_matches_numpy(ctx):
    """rails_panel_gemm_wide: any number of output columns from one upload of C (the basis rotation P <- P Q)"""
    import ctypes as C

    from rails_amd.wrappers import _p

    g = np.random.default_rng(8)
    m, k, r = 3001, 75, 300
    Xh = g.uniform(-1, 1, (m, k))
    Cm = np.asfortranarray(np.pad(g.uniform(-1, 1, (k, r)), ((0, 5), (0, 0))))  # ldc = k + 5
    Yh0 = g.uniform(-1, 1, (m, r))
    X = MV(ctx, data=Xh)
    Y = MV(ctx, m=m, n=r + 3, capacity=r + 3)
    for alpha, beta in ((1.0, 0.0), (-0.5, 1.0)):
        Yv = Y.view(2, r + 1)
        Yv.from_host(Yh0)
        rc = ctx.lib.rails_panel_gemm_wide(ctx.h, alpha, X.panel.h, 0, k, _p(Cm), k + 5, r, beta, Y.panel.h, 2)
        assert rc == 0
        want = beta * Yh0 + alpha * (Xh @ Cm[:k])
        np.testing.assert_allclose(Yv.to_host(), want, rtol=0, atol=1e-13 * k)
    # overlapping windows of one panel are refused
    assert ctx.lib.rails_panel_gemm_wide(ctx.h, 1.0, Y.panel.h, 0, 10, _p(Cm), k + 5, 20, 0.0, Y.panel.h, 5) != 0


def test_busy_meter_counts_device_time(ctx):
    """rails_ctx_set_meter: with the meter on, every launch is bracketed by events and `gpu_busy_ms` adds up the device time -- positive,
    below the wall-clock time of the same work, and unchanged while the meter is off."""
    import time

    import rails_amd

    X = rails_amd.HipMultiVectorWrapper(ctx, m=400000, n=64, capacity=64)
    X.random()
    ctx.sync()
    base = ctx.stats()["gpu_busy_ms"]
    G = X.dot(X)
    ctx.sync()
    assert ctx.stats()["gpu_busy_ms"] == base  # meter off: nothing is counted
    ctx.set_meter(True)
    t0 = time.perf_counter()
    for _ in range(5):
        G = X.dot(X)
    ctx.sync()
    wall_ms = 1e3 * (time.perf_counter() - t0)
    busy = ctx.stats()["gpu_busy_ms"] - base
    ctx.set_meter(False)
    assert 0.0 < busy <= wall_ms, (busy, wall_ms)
    assert busy > 5 * 0.02  # five Gram launches over 200 MB each take longer than 20 us apiece
    after = ctx.stats()["gpu_busy_ms"]
    G2 = X.dot(X)
    ctx.sync()
    assert ctx.stats()["gpu_busy_ms"] == after
    np.testing.assert_allclose(G2, G, rtol=0, atol=0)


def test_deferred_small_results(ctx):
    """rails_gram_deferred / rails_panel_gemm_deferred / rails_chol_inverse_deferred (include/rails_hip.h): the chain the coordinate-space
    back end queues behind the host's projected solve -- Gram into a device slot, an update whose coefficients come from that slot, the
    scaled Cholesky factor of a block's Gram matrix inverted on the device -- against numpy, with ONE synchronisation at the end."""
    import ctypes as C

    import rails_amd

    lib = ctx.lib
    g = np.random.default_rng(21)
    m, a, w = 50000, 24, 17
    Xh = g.uniform(-1, 1, (m, a + w))
    # make the last w columns nearly lie in the span of the first a: what a projection round sees
    Xh[:, a:] = Xh[:, :a] @ g.uniform(-1, 1, (a, w)) + 0.05 * g.uniform(-1, 1, (m, w))
    X = rails_amd.HipMultiVectorWrapper(ctx, data=Xh)
    P = X.panel.h
    rails_amd._lib.check(lib.rails_deferred_reserve(ctx.h, 5, (a + w) * w), "rails_deferred_reserve")
    chk = rails_amd._lib.check
    chk(lib.rails_gram_deferred(ctx.h, P, 0, a + w, P, a, w, 0), "gram")                 # slot 0: [Xa | Xw]' Xw, leading dimension a + w
    chk(lib.rails_panel_gemm_deferred(ctx.h, -1.0, P, 0, a, 0, a + w, w, 1.0, P, a), "update")  # Xw -= Xa * (top a rows of slot 0)
    chk(lib.rails_gram_deferred(ctx.h, P, a, w, P, a, w, 1), "gram")                      # slot 1: Gram matrix of the updated block
    chk(lib.rails_chol_inverse_deferred(ctx.h, 1, w, 2), "chol")                          # slot 2: D^-1 R^-1
    chk(lib.rails_panel_gemm_deferred(ctx.h, 1.0, P, a, w, 2, w, w, 0.0, P, a), "scale")    # Xw <- Xw * slot 2 (in place)
    chk(lib.rails_gram_deferred(ctx.h, P, a, w, P, a, w, 3), "gram")                      # slot 3: should be the identity
    ctx.sync()
    dp = C.POINTER(C.c_double)

    def fetch(slot, rows, cols):
        out = np.zeros((rows, cols), order="F")
        chk(lib.rails_deferred_fetch(ctx.h, slot, rows * cols, out.ctypes.data_as(dp)), "fetch")
        return out

    C0 = fetch(0, a + w, w)
    np.testing.assert_allclose(C0, Xh.T @ Xh[:, a:], rtol=0, atol=1e-9 * m)
    Xw1 = Xh[:, a:] - Xh[:, :a] @ C0[:a]
    G1 = fetch(1, w, w)
    np.testing.assert_allclose(G1, Xw1.T @ Xw1, rtol=1e-9, atol=1e-9 * np.abs(G1).max())
    M = fetch(2, w, w)
    d = np.sqrt(np.diag(G1))
    R = np.linalg.cholesky(G1 / np.outer(d, d)).T
    np.testing.assert_allclose(M, np.triu(np.linalg.inv(R) / d[:, None]), rtol=1e-9, atol=1e-12 * np.abs(M).max())
    np.testing.assert_allclose(fetch(3, w, w), np.eye(w), rtol=0, atol=1e-10)
    np.testing.assert_allclose(X.to_host()[:, a:], Xw1 @ M, rtol=0, atol=1e-10 * np.abs(Xw1 @ M).max())


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,r,r2,xoff", [(50000, 100, 17, 16, 0), (60007, 352, 17, 16, 0), (40000, 300, 16, 16, 2), (30000, 420, 17, 12, 0),
                                           (20000, 64, 5, 3, 0), (50000, 200, 17, 16, 1), (50000, 120, 20, 18, 0), (3000, 90, 17, 16, 0),
                                           (45001, 353, 17, 16, 0), (33000, 130, 17, 16, 0), (33003, 410, 17, 16, 4), (4100, 128, 1, 1, 0), (70000, 255, 9, 9, 6)])
def test_update_and_second_projection_in_one_pass(ctx, m, k, r, r2, xoff):
    """rails_update_gram_deferred (include/rails_hip.h): Y += alpha X C, then slot <- X' Y[:, :r2], as one pass over X (k_update_gram:
    16 MFMA columns + the 17th on the vector unit; odd window offsets, 20 columns or few rows take the two separate kernels) against numpy.  The second sweep of the reference's block Gram-Schmidt (src/StlWrapper.cpp:314-344)."""
    import ctypes as C

    import rails_amd

    lib = ctx.lib
    chk = rails_amd._lib.check
    g = np.random.default_rng(m + k)
    Xh = g.uniform(-1, 1, (m, xoff + k + r + 3))
    X = rails_amd.HipMultiVectorWrapper(ctx, data=Xh)
    P = X.panel.h
    Ch = np.asfortranarray(g.uniform(-1, 1, (k + 5, r)))  # leading dimension k + 5
    dp = C.POINTER(C.c_double)
    chk(lib.rails_deferred_reserve(ctx.h, 3, (k + 64) * 32), "rails_deferred_reserve")
    before = ctx.stats().get("update_gram_fused", 0)
    chk(lib.rails_update_gram_deferred(ctx.h, -0.5, P, xoff, k, Ch.ctypes.data_as(dp), k + 5, r, P, xoff + k, r2, 1), "rails_update_gram_deferred")
    ctx.sync()
    fused = ctx.stats().get("update_gram_fused", 0) - before
    assert fused == (1 if (r <= 17 and m >= 4096 and xoff % 2 == 0) else 0)
    Xa = Xh[:, xoff:xoff + k]
    Ynew = Xh[:, xoff + k:xoff + k + r] - 0.5 * (Xa @ Ch[:k])
    out = X.to_host()
    scale = np.abs(Ynew).max()
    np.testing.assert_allclose(out[:, xoff + k:xoff + k + r], Ynew, rtol=0, atol=1e-13 * k * scale)
    assert np.array_equal(out[:, :xoff + k], Xh[:, :xoff + k]) and np.array_equal(out[:, xoff + k + r:], Xh[:, xoff + k + r:])
    C2 = np.zeros((k, r2), order="F")
    chk(lib.rails_deferred_fetch(ctx.h, 1, k * r2, C2.ctypes.data_as(dp)), "fetch")
    ref = Xa.T @ Ynew[:, :r2]
    np.testing.assert_allclose(C2, ref, rtol=0, atol=1e-13 * m * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("w", [1, 2, 16, 17, 31, 32, 33, 48])
def test_small_cholesky_inverse_on_the_device(ctx, w):
    """rails_chol_inverse_deferred (k_small_chol: one wave, a column per lane): M = D^-1 R^-1 with R'R = D^-1 G D^-1, so that M' G M = I,
    for every block width the back end uses, on a Gram matrix with a wide range of column norms."""
    import ctypes as C

    import rails_amd

    lib = ctx.lib
    chk = rails_amd._lib.check
    g = np.random.default_rng(w)
    m = 20000
    Xh = g.uniform(-1, 1, (m, w)) * np.logspace(0, 5, w)[None, :]
    X = rails_amd.HipMultiVectorWrapper(ctx, data=Xh)
    chk(lib.rails_deferred_reserve(ctx.h, 3, 64 * 48), "rails_deferred_reserve")
    chk(lib.rails_gram_deferred(ctx.h, X.panel.h, 0, w, X.panel.h, 0, w, 0), "gram")
    chk(lib.rails_chol_inverse_deferred(ctx.h, 0, w, 1), "chol")
    ctx.sync()
    dp = C.POINTER(C.c_double)
    G = np.zeros((w, w), order="F")
    M = np.zeros((w, w), order="F")
    chk(lib.rails_deferred_fetch(ctx.h, 0, w * w, G.ctypes.data_as(dp)), "fetch")
    chk(lib.rails_deferred_fetch(ctx.h, 1, w * w, M.ctypes.data_as(dp)), "fetch")
    assert np.array_equal(M, np.triu(M))
    np.testing.assert_allclose(M.T @ G @ M, np.eye(w), rtol=0, atol=1e-11)
    d = np.sqrt(np.diag(G))
    R = np.linalg.cholesky(G / np.outer(d, d)).T
    np.testing.assert_allclose(M, np.triu(np.linalg.inv(R) / d[:, None]), rtol=1e-8, atol=1e-12 * np.abs(M).max())


@pytest.mark.gpu
@pytest.mark.parametrize("nc,xoff,yoff", [(16, 0, 0), (16, 2, 3), (12, 0, 0), (11, 0, 2), (9, 4, 1), (16, 1, 0), (13, 3, 2), (32, 0, 0), (24, 2, 1), (31, 0, 0)])
def test_narrow_product_is_bitwise_the_rowgather_result(ctx, oracle, nc, xoff, yoff):
    """The in-loop product A * W at Expand size <= 32 (`A_ * W`, src/LyapunovSolver.hpp:146; k_spmm_narrow, spmm.hip kernel 1c): rows of
    0 to 60 entries (full groups of eight, tails of one to seven, blocks whose entries do not fit the LDS buffer), odd last columns,
    odd output offsets and odd input offsets (rows that are only 8-byte aligned: W as a view of V in the direct back end), against the whole-width row-gather kernel bit for bit and against the oracle's CSR product."""
    import rails_amd

    g = np.random.default_rng(nc + xoff)
    m = 20000
    lens = g.integers(0, 28, m)
    lens[5000:5200] = 60  # 64 consecutive rows x 60 entries: more than the 2048 the kernel stages
    lens[::97] = 0
    rowptr = np.zeros(m + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    col = np.concatenate([np.sort(g.choice(m, n, replace=False)) for n in lens]).astype(np.int32)
    val = g.uniform(-1, 1, col.size)
    op = rails_amd.HipOperatorWrapper(ctx, rowptr, col, val)
    Xh = g.uniform(-1, 1, (m, nc))
    big = MV(ctx, m=m, n=nc + xoff, capacity=nc + xoff + 1)
    X = big.view(xoff, xoff + nc - 1)
    X.from_host(Xh)
    outp = MV(ctx, m=m, n=nc + yoff, capacity=nc + yoff + 3)
    Y = outp.view(yoff, yoff + nc - 1)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_narrow"
    Yn = Y.to_host()
    op.set_variant(3)
    Y3 = op.apply(X)
    assert op.last_kernel() == "k_spmm_rowgather"
    assert np.array_equal(Y3.to_host(), Yn)
    ref = oracle.csr_spmm(rowptr, col, val, Xh)
    assert np.abs(Yn - ref).max() <= 1e-14 * np.sqrt(60) * 4 * np.abs(ref).max()
