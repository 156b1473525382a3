"""The C-ABI library loads on a CPU-only host and exports every symbol the public headers declare
(no device compute is attempted here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rails_[a-z0-9_]+)\s*\(", text))
    typedefs = set(re.findall(r"\(\*\s*(rails_[a-z0-9_]+)\s*\)", text))
    return sorted(names - typedefs)


def test_library_exports_every_declared_symbol():
    import rails_amd

    lib = rails_amd.load()
    headers = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert "rails_hip.h" in headers
    missing = []
    for h in headers:
        for name in _declared(h):
            if not hasattr(lib, name):
                missing.append((h, name))
    assert not missing, missing
    assert b"gfx950" in lib.rails_version()


def test_no_cpu_fallback_without_gpu():
    """On a host without a gfx950 device the context creation must fail loudly, not fall back."""
    import torch

    import rails_amd

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(rails_amd.RailsError):
        rails_amd.Context(device=0)


def test_host_sb03md_kats():
    # test/SlicotWrapper_test.cpp:7-38 through the product's host C ABI
    import rails_amd

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)
    A = np.array([[2.0]], order="F")
    X = np.array([[-4.0]], order="F")
    scale, info = C.c_double(1.0), C.c_int(0)
    lib.rails_sb03md(b"C", b"X", b"N", b"T", 1, A.ctypes.data_as(dp), 1, X.ctypes.data_as(dp), 1, C.byref(scale), C.byref(info))
    assert X[0, 0] == -1.0 and info.value == 0
    A = np.array([[0.0, 1.0], [-5.0, -5.0]], order="F")
    X = np.array([[-1.0, 0.0], [0.0, -1.0]], order="F")
    lib.rails_sb03md(b"C", b"X", b"N", b"T", 2, A.ctypes.data_as(dp), 2, X.ctypes.data_as(dp), 2, C.byref(scale), C.byref(info))
    np.testing.assert_allclose(X, [[0.62, -0.5], [-0.5, 0.6]], rtol=0, atol=1e-14)
    assert info.value == 0 and scale.value == 1.0


def test_host_sb03md_matches_oracle(oracle):
    import rails_amd

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)
    g = np.random.default_rng(0)
    for n in (3, 17, 64):
        A = g.uniform(-1, 1, (n, n)) - 4 * np.eye(n)
        Cm = g.uniform(-1, 1, (n, n))
        Cm = Cm + Cm.T
        Xo, sc, info = oracle.sb03md(A, Cm)
        Ap, Xp = np.asfortranarray(A.copy()), np.asfortranarray(Cm.copy())
        scale, inf = C.c_double(1.0), C.c_int(0)
        lib.rails_sb03md(b"C", b"X", b"N", b"T", n, Ap.ctypes.data_as(dp), n, Xp.ctypes.data_as(dp), n, C.byref(scale), C.byref(inf))
        assert inf.value == 0
        np.testing.assert_allclose(Xp, Xo, atol=1e-12)
        assert np.abs(A @ Xp + Xp @ A.T - scale.value * Cm).max() < 1e-12
    # symmetric A takes the eigen-decomposition path (and n >= 64 the blocked triangular solver on the general path): the same
    # equation, checked against the oracle's Bartels-Stewart solve and by its residual, for both values of trans
    for n in (9, 40, 130):
        S = g.uniform(-1, 1, (n, n))
        A = -(S @ S.T) / n - np.eye(n)
        Cm = g.uniform(-1, 1, (n, n))
        Cm = Cm + Cm.T
        Xo, sc, info = oracle.sb03md(A, Cm)
        for trans in (b"T", b"N"):
            Ap, Xp = np.asfortranarray(A.copy()), np.asfortranarray(Cm.copy())
            scale, inf = C.c_double(1.0), C.c_int(0)
            lib.rails_sb03md(b"C", b"X", b"N", trans, n, Ap.ctypes.data_as(dp), n, Xp.ctypes.data_as(dp), n, C.byref(scale), C.byref(inf))
            assert inf.value == 0 and scale.value == 1.0
            np.testing.assert_allclose(Xp, Xo, atol=1e-11)
            assert np.abs(A @ Xp + Xp @ A.T - Cm).max() < 1e-11
            assert np.abs(Ap - np.diag(np.diag(Ap))).max() == 0.0  # A comes back as its (diagonal) Schur form
    n = 130
    A = g.uniform(-1, 1, (n, n)) - 6 * np.eye(n)
    Cm = g.uniform(-1, 1, (n, n))
    Cm = Cm + Cm.T
    Xo, sc, info = oracle.sb03md(A, Cm)
    Ap, Xp = np.asfortranarray(A.copy()), np.asfortranarray(Cm.copy())
    scale, inf = C.c_double(1.0), C.c_int(0)
    lib.rails_sb03md(b"C", b"X", b"N", b"T", n, Ap.ctypes.data_as(dp), n, Xp.ctypes.data_as(dp), n, C.byref(scale), C.byref(inf))
    assert inf.value == 0
    np.testing.assert_allclose(Xp * sc / scale.value, Xo, atol=1e-11)


def test_host_dsyev_and_dsteqr():
    import rails_amd

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)
    # test/GenericDenseMatrixWrapper_test.cpp eig KAT shape: symmetric tridiagonal, ascending eigenvalues
    n = 6
    d = np.arange(1.0, n + 1)
    e = np.full(n - 1, 0.5)
    S = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    a = np.asfortranarray(S.copy())
    w = np.zeros(n)
    info = C.c_int(0)
    lib.rails_dsyev(b"V", b"U", n, a.ctypes.data_as(dp), n, w.ctypes.data_as(dp), C.byref(info))
    assert info.value == 0
    np.testing.assert_allclose(w, np.linalg.eigvalsh(S), atol=1e-13)
    np.testing.assert_allclose(a @ np.diag(w) @ a.T, S, atol=1e-13)
    z = np.asfortranarray(np.eye(n))
    dd, ee, work = d.copy(), e.copy(), np.zeros(2 * n)
    lib.rails_dsteqr(b"I", n, dd.ctypes.data_as(dp), ee.ctypes.data_as(dp), z.ctypes.data_as(dp), n, work.ctypes.data_as(dp), C.byref(info))
    assert info.value == 0
    np.testing.assert_allclose(dd, w, atol=1e-13)


def test_host_sb03md_smith_route_is_verified_and_falls_back():
    """rails_sb03md on nonsymmetric matrices of 32 rows and more tries the squared Smith iteration first (level-3 BLAS, residual-verified)
    and runs Bartels-Stewart where that does not apply: both routes against scipy's Bartels-Stewart, both `trans` forms"""
    import scipy.linalg as sl

    import rails_amd

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)

    def counts():
        a, b = C.c_long(0), C.c_long(0)
        lib.rails_sb03md_counts(C.byref(a), C.byref(b))
        return a.value, b.value

    def solve(A, Cm, trans):
        n = A.shape[0]
        Ap, X = np.asfortranarray(A.copy()), np.asfortranarray(Cm.copy())
        scale, info = C.c_double(0), C.c_int(0)
        lib.rails_sb03md(b"C", b"X", b"N", trans, n, Ap.ctypes.data_as(dp), n, X.ctypes.data_as(dp), n, C.byref(scale), C.byref(info))
        assert info.value == 0 and scale.value == 1.0
        return X

    g = np.random.default_rng(7)
    # (a) clustered spectrum (the projection of a diagonally dominant operator): the Smith route, to rounding
    for n in (32, 75, 160):
        A = -14.0 * np.eye(n) + 0.6 * g.uniform(0, 1, (n, n)) * (g.uniform(size=(n, n)) < 27.0 / n)
        Bm = g.standard_normal((n, 5))
        Cm = -(Bm @ Bm.T)
        for trans in (b"T", b"N"):
            lib.rails_sb03md_set_pause(0)
            s0, b0 = counts()
            X = solve(A, Cm, trans)
            assert counts() == (s0 + 1, b0)
            Mm = A if trans == b"T" else A.T
            Xref = sl.solve_continuous_lyapunov(Mm, Cm)
            assert np.linalg.norm(X - Xref) <= 5e-14 * np.linalg.norm(Xref)
            assert np.linalg.norm(Mm @ X + X @ Mm.T - Cm) <= 1e-14 * (2 * np.linalg.norm(Mm) * np.linalg.norm(X) + np.linalg.norm(Cm))
            assert np.array_equal(X, X.T)
    # (b) a spectrum spread over three decades, stable: the factored ADI form (several shifts) takes it; no single shift would
    n = 96
    spread = -np.diag(np.logspace(-1, 2, n)) + 0.02 * g.standard_normal((n, n))
    assert np.linalg.eigvals(spread).real.max() < 0
    Bm = g.standard_normal((n, 3))
    Cm = -(Bm @ Bm.T)
    lib.rails_sb03md_set_pause(0)
    s0, b0 = counts()
    X = solve(spread, Cm, b"T")
    assert counts() == (s0 + 1, b0)
    Xref = sl.solve_continuous_lyapunov(spread, Cm)
    assert np.linalg.norm(X - Xref) <= 1e-11 * np.linalg.norm(Xref)
    assert np.linalg.norm(spread @ X + X @ spread.T - Cm) <= 1e-14 * (2 * np.linalg.norm(spread) * np.linalg.norm(X) + np.linalg.norm(Cm))
    # (c) an unstable matrix, (d) a full-rank indefinite right-hand side on a spread spectrum: Bartels-Stewart answers
    unstable = spread.copy()
    unstable[0, 0] = 3.0
    Cfull = g.standard_normal((n, n))
    Cfull = Cfull + Cfull.T
    wide = -np.diag(np.logspace(-3, 2, n)) + 1e-4 * g.standard_normal((n, n))
    for A, Cc in ((unstable, Cm), (wide, Cfull)):
        lib.rails_sb03md_set_pause(0)
        s0, b0 = counts()
        X = solve(A, Cc, b"T")
        s1, b1 = counts()
        assert s1 == s0 and b1 == b0 + 1
        Xref = sl.solve_continuous_lyapunov(A, Cc)
        assert np.linalg.norm(X - Xref) <= 1e-8 * np.linalg.norm(Xref)  # conditioning of these problems, same algorithm on both sides


def test_host_sb03md_builds_on_the_call_before_inside_a_restart_cycle():
    """Inside a restart cycle the solver's projected matrix grows by bordering (V'AV keeps its leading block, src/LyapunovSolver.hpp:146-160):
    the factored ADI route of rails_sb03md then keeps its shifts and EXTENDS the inverses / LU factors of M - p I of the call before instead
    of recomputing them (host_numerics.cpp, AdiCache).  The same projected equations, solved in the order a run meets them, have to come
    out as Bartels-Stewart gives them whether a call inherited or not; a border that moves the spectrum, an unrelated matrix and a smaller
    one must not be served from the cache."""
    import scipy.linalg as sl

    import rails_amd

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)

    def counts():
        a, b = C.c_long(0), C.c_long(0)
        lib.rails_sb03md_adi_counts(C.byref(a), C.byref(b))
        return a.value, b.value

    def solve(A, Bm, trans=b"T"):
        n = A.shape[0]
        Ap, X = np.asfortranarray(A.copy()), np.asfortranarray(-(Bm @ Bm.T))
        scale, info = C.c_double(0), C.c_int(0)
        lib.rails_sb03md(b"C", b"X", b"N", trans, n, Ap.ctypes.data_as(dp), n, X.ctypes.data_as(dp), n, C.byref(scale), C.byref(info))
        assert info.value == 0 and scale.value == 1.0
        ref = sl.solve_continuous_lyapunov(A if trans == b"T" else A.T, -(Bm @ Bm.T))  # 'T': A X + X A' = C, 'N': A' X + X A = C
        return np.abs(X - ref).max() / np.abs(ref).max()

    g = np.random.default_rng(3)
    N = 208
    Afull = -3.0 * np.eye(N) + 0.8 * g.standard_normal((N, N)) / np.sqrt(N)
    Bfull = g.standard_normal((N, 16))
    lib.rails_sb03md_set_pause(0)
    e0, f0 = counts()
    sizes = [48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208]  # 48 -> 64: from the LU form to the explicit step operators (a fresh start)
    for n in sizes:
        assert solve(Afull[:n, :n], Bfull[:n]) <= 1e-11, n
    e1, f1 = counts()
    assert e1 - e0 >= len(sizes) - 3 and f1 - f0 >= 2  # (the first call, and the first one of the explicit form, start from scratch)
    # the same matrix again (a second right-hand side): served from the cache
    assert solve(Afull, g.standard_normal((N, 16))) <= 1e-11
    assert counts()[0] == e1 + 1
    # a border that widens the spectrum by two orders of magnitude: the inherited shifts do not cover it, the call starts from scratch
    Aw = np.zeros((N + 16, N + 16))
    Aw[:N, :N] = Afull
    Aw[N:, N:] = -0.03 * np.eye(16)
    Aw[:N, N:] = 0.01 * g.standard_normal((N, 16))
    Aw[N:, :N] = 0.01 * g.standard_normal((16, N))
    e2, f2 = counts()
    assert solve(Aw, g.standard_normal((N + 16, 16))) <= 1e-10
    assert counts() == (e2, f2 + 1)
    # an unrelated matrix of the next size, and a smaller matrix: nothing to build on
    A2 = -2.0 * np.eye(N + 32) + 0.5 * g.standard_normal((N + 32, N + 32)) / np.sqrt(N + 32)
    assert solve(A2, g.standard_normal((N + 32, 16))) <= 1e-11
    assert solve(A2[:100, :100], g.standard_normal((100, 16))) <= 1e-11
    assert counts() == (e2, f2 + 3)
    # the other form of the equation (A' X + X A = C) on its own bordered sequence: the cache is kept per form
    e3, f3 = counts()
    for n in (96, 112, 128, 144):
        assert solve(Afull[:n, :n], Bfull[:n], trans=b"N") <= 1e-11, n
    e4, f4 = counts()
    assert e4 - e3 == 3 and f4 - f3 == 1


def test_host_sb03md_at_the_c4_size():
    """The projected equation at BASELINE configs[3]'s size (Restart size 256, B m x 32): V'AV of a nonsymmetric 27-point stencil operator
    on a 256-dimensional block Krylov space, right-hand side -(V'B)(V'B)' of rank 32.  Bartels-Stewart costs 21 ms there
    (profiles/r01_host_lyap.txt); the factored ADI route has to take it (route counters, as bench.py reports them), verified against
    scipy's Bartels-Stewart.  The symmetric stencil of C4 itself takes the eigen-decomposition path: checked beside it."""
    import time

    import scipy.linalg as sl
    import scipy.sparse as sp

    import rails_amd
    from rails_amd import problems as P

    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)

    def counts():
        a, b = C.c_long(0), C.c_long(0)
        lib.rails_sb03md_counts(C.byref(a), C.byref(b))
        return a.value, b.value

    def projected(random_values):
        rowptr, col, val = P.stencil27(24, 24, 12, random_values=random_values, seed=5)
        m = rowptr.size - 1
        A = sp.csr_matrix((val, col, rowptr), shape=(m, m))
        V = np.linalg.qr(P.rhs(m, 32, seed=9))[0]
        blocks = [V]
        while sum(b.shape[1] for b in blocks) < 256:  # block Krylov space, orthonormalised block by block (twice)
            W = A @ blocks[-1]
            Q = np.hstack(blocks)
            W -= Q @ (Q.T @ W)
            W -= Q @ (Q.T @ W)
            blocks.append(np.linalg.qr(W)[0])
        V = np.hstack(blocks)[:, :256]
        Bv = V.T @ P.rhs(m, 32, seed=9)
        return V.T @ (A @ V), -(Bv @ Bv.T)

    for random_values, want_route in ((True, "adi"), (False, "symmetric")):
        M, Cm = projected(random_values)
        n = M.shape[0]
        assert n == 256
        lib.rails_sb03md_set_pause(0)
        s0, b0 = counts()
        Ap, X = np.asfortranarray(M.copy()), np.asfortranarray(Cm.copy())
        scale, info = C.c_double(0), C.c_int(0)
        t = time.perf_counter()
        lib.rails_sb03md(b"C", b"X", b"N", b"T", n, Ap.ctypes.data_as(dp), n, X.ctypes.data_as(dp), n, C.byref(scale), C.byref(info))
        dt = time.perf_counter() - t
        s1, b1 = counts()
        assert info.value == 0 and scale.value == 1.0
        if want_route == "adi":
            assert (s1, b1) == (s0 + 1, b0), "the nonsymmetric n = 256 projection fell back to Bartels-Stewart"
        else:
            assert (s1, b1) == (s0, b0 + 1) or (s1, b1) == (s0, b0)  # eigen-decomposition path (counted with the direct solves, if at all)
        Xref = sl.solve_continuous_lyapunov(M, Cm)
        assert np.linalg.norm(X - Xref) <= 1e-10 * np.linalg.norm(Xref)
        assert np.linalg.norm(M @ X + X @ M.T - Cm) <= 1e-13 * (2 * np.linalg.norm(M) * np.linalg.norm(X) + np.linalg.norm(Cm))
        print("n = 256, %s stencil: %s route, %.1f ms" % ("random" if random_values else "fixed", want_route, 1e3 * dt))


@pytest.mark.parametrize("loops", [0, 1])
def test_host_dtrsm_all_forms(loops, monkeypatch):
    """rails_dtrsm (the six triangular solves of the generalized projected solve): BLAS path and the loop fallback, every
    side / uplo / trans / diag combination, leading dimensions larger than the sizes"""
    import scipy.linalg as sl

    import rails_amd

    monkeypatch.setenv("RAILS_DTRSM_LOOPS", str(loops))
    lib = rails_amd.load()
    dp = C.POINTER(C.c_double)
    g = np.random.default_rng(3)
    m, n = 7, 5
    for side in (b"L", b"R"):
        na = m if side == b"L" else n
        for uplo in (b"L", b"U"):
            for trans in (b"N", b"T"):
                for diag in (b"N", b"U"):
                    A = g.uniform(-1, 1, (na, na)) + 4 * np.eye(na)
                    Af = np.asfortranarray(np.pad(A, ((0, 3), (0, 0))))          # lda = na + 3; the other triangle holds junk
                    Bm = g.uniform(-1, 1, (m, n))
                    Bf = np.asfortranarray(np.pad(Bm, ((0, 2), (0, 0))))         # ldb = m + 2
                    lib.rails_dtrsm(side, uplo, trans, diag, m, n, 0.5, Af.ctypes.data_as(dp), na + 3, Bf.ctypes.data_as(dp), m + 2)
                    Tm = np.tril(A) if uplo == b"L" else np.triu(A)
                    if diag == b"U":
                        np.fill_diagonal(Tm, 1.0)
                    op = Tm.T if trans == b"T" else Tm
                    want = 0.5 * (np.linalg.solve(op, Bm) if side == b"L" else np.linalg.solve(op.T, Bm.T).T)
                    np.testing.assert_allclose(Bf[:m], want, rtol=0, atol=1e-13, err_msg=str((side, uplo, trans, diag)))
                    assert np.array_equal(Bf[m:], np.zeros((2, n)))


def test_host_pivoted_cholesky_reveals_rank():
    # rails_dpstrf (projected-space residual Lanczos, rails/HipSolverOps.hpp): P'SP = R'R on a rank-deficient Gram matrix
    import rails_amd

    lib = rails_amd.load()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rng = np.random.default_rng(3)
    n, r = 12, 7
    W = rng.standard_normal((40, r)) @ rng.standard_normal((r, n))
    S = W.T @ W
    a = np.asfortranarray(S.copy())
    piv = np.zeros(n, dtype=np.int32)
    rank, info = C.c_int(0), C.c_int(0)
    lib.rails_dpstrf(b"U", n, a.ctypes.data_as(dp), n, piv.ctypes.data_as(ip), C.byref(rank), 1e-10 * S.diagonal().max(), C.byref(info))
    assert rank.value == r and info.value == 1
    R = np.triu(a)[:r, :]
    np.testing.assert_allclose(R.T @ R, S[np.ix_(piv, piv)], atol=1e-9 * S.max())


def test_wrappers_instantiate_the_reference_solver_template():
    """Compile-only (no link, no run): the reference's RAILS::Solver template accepts the HIP wrapper classes."""
    import subprocess

    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference is only present in the build container")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-I/root/reference", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "rails_amd", "include"), os.path.join(ROOT, "tests", "integration", "reference_solver_instantiation.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shipped_library_has_no_experiment_builds_of_the_sweep_kernel():
    """The sweep kernel's experiment instantiations (results wrong by construction: timings only) are compiled only with
    `make EXPERIMENTS=1`; the shipped library holds the two product kernels and refuses RAILS_SWEEP_ABLATE / RAILS_SWEEP_LAYOUT."""
    import rails_amd._lib as L

    blob = open(L.LIB_PATH, "rb").read()
    assert b"k_spmm_sweep_h2" in blob
    for name in (b"k_spmm_sweep_noread", b"k_spmm_sweep_nofma", b"k_spmm_sweep_nowait", b"k_spmm_sweep_noidx", b"k_spmm_sweep_bare",
                 b"k_spmm_sweep_nodpp", b"k_spmm_sweep_halves", b"k_spmm_sweep_switches"):
        assert name not in blob, name
    assert b"no experiment builds of the sweep kernel" in blob  # the refusal's message
