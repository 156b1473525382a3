"""The RAILS solver on the HIP backend (through the C ABI of include/rails_solver.h) against the reference's
known-answer tests and against the CPU oracle on identical seeded inputs (same counter-based RNG streams).

Parity tolerances (fp64): Lanczos residual estimates agree to rel. 1e-6 over the first
trips (before rounding-level differences are amplified by the Krylov recurrences), same trip count +-1,
||V T V'_gpu - V T V'_cpu||_F / ||.||_F <= 10*tol, both satisfy the reference's own acceptance
(|R|_max <= 1e-3 on the small dense cases, test/TestHelpers.hpp:10)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=1)
    yield c
    c.close()


def _solve(ctx, A, B, params, seed=1, M=None, V0=None, mass=False, options=None):
    import rails_amd
    from rails_amd import problems as P

    ctx.set_seed(seed, 0)
    csr = A if isinstance(A, tuple) else P.dense_to_csr(A)
    op = rails_amd.HipOperatorWrapper(ctx, *csr)
    mop = rails_amd.HipOperatorWrapper(ctx, *M) if M is not None else None
    s = rails_amd.Solver(ctx, op, B, M=mop)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    if mass:
        s.set_option("mass", 1)
    opts = {"subspace": 0}  # these tests address the direct back end unless a test says otherwise (SUB below)
    opts.update(options or {})
    for name, value in opts.items():
        s.set_option(name, value)
    code, V, T = s.solve(V0=V0)
    return code, V, T, s


def _residual(A, B, V, T):
    X = V @ T @ V.T
    return A @ X + X @ A.T + B @ B.T


def test_solver_2x2_kats(ctx):
    # test/LyapunovSolverEpetra_test.cpp:51-106,109-177
    A = np.array([[0.0, 1.0], [-5.0, -5.0]])
    for B, Xexp in ((-np.eye(2), [[0.62, -0.5], [-0.5, 0.6]]), (np.array([[-1.0], [-1.0]]), [[0.82, -0.5], [-0.5, 0.6]])):
        code, V, T, s = _solve(ctx, A, B, {"Minimize solution space": 0})
        assert code == 0
        np.testing.assert_allclose(V @ T @ V.T, Xexp, rtol=0, atol=1e-13)


def test_set_parameters_rejects_lanczos_le_expand(ctx):
    # src/LyapunovSolver.hpp:89-95
    import rails_amd
    from rails_amd import problems as P

    op = rails_amd.HipOperatorWrapper(ctx, *P.laplace7(3, 3, 3))
    s = rails_amd.Solver(ctx, op, P.rhs(27, 1))
    assert s.set_parameters({"Lanczos iterations": 3, "Expand size": 3}) == 1
    assert s.set_parameters({"LANCZOS ITERATIONS": 10, "expand size": 3}) == 0  # case variants (:40-70)


def _tridiagonal_problem(n, seed):
    g = np.random.default_rng(seed)
    A = g.uniform(-1, 1, (n, n))
    for i in range(n):
        for j in range(n):
            if abs(i - j) > 1:
                A[i, j] = 0.0
            elif i == j:
                A[i, j] *= 3.0
    B = np.zeros((n, 1))
    B[n - 1, 0] = g.uniform(-1, 1)
    return A, B


def test_solver_n20_reference_cases(ctx):
    # test/LyapunovSolver_test.cpp:118-352 (dense, restart, minimise, restart iterations, warm start)
    g = np.random.default_rng(1)
    n = 20
    A = g.uniform(-1, 1, (n, n))
    B = np.zeros((n, 1))
    B[n - 1, 0] = g.uniform(-1, 1)
    code, V, T, _ = _solve(ctx, A, B, {})
    assert code == 0 and np.abs(_residual(A, B, V, T)).max() < 1e-3

    A, B = _tridiagonal_problem(20, 2)
    code, V, T, _ = _solve(ctx, A, B, {"Restart Size": 19, "Reduced Size": 15, "Expand Size": 1, "Minimize solution space": 0})
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3

    A, B = _tridiagonal_problem(20, 3)
    code, V, T, _ = _solve(ctx, A, B, {"Minimize solution space": 0, "Tolerance": 1e-8})
    assert code == 0 and V.shape[1] == 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3
    code, V, T, _ = _solve(ctx, A, B, {"Minimize solution space": 1, "Tolerance": 1e-8})
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3

    A, B = _tridiagonal_problem(20, 4)
    code, V, T, _ = _solve(ctx, A, B, {"Restart iterations": 10, "Minimize solution space": 0, "Expand size": 1})
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3

    A, B = _tridiagonal_problem(20, 5)
    prm = {"Minimize solution space": 1, "Tolerance": 1e-8}
    code, V, T, _ = _solve(ctx, A, B, prm)
    assert code == 0 and V.shape[1] < 20
    A[19, 19] = 4.0
    code, V2, T2, _ = _solve(ctx, A, B, {**prm, "Restart from solution": 1}, V0=V)
    assert code == 0 and V2.shape[1] < 20 and np.abs(_residual(A, B, V2, T2)).max() < 1e-3


def _compare_with_oracle(ctx, oracle, A, B, params, tol, seed, M=None, nhist=6, trajectory=True, options=None, first_rtol=1e-9):
    """trajectory=True needs `Lanczos iterations` <= 2 + p: otherwise the first trips run Lanczos past the rank of the
    residual operator and the expansion vectors are rounding-level chaotic in the reference algorithm itself
    (oracle/README.md); then only invariants are compared."""
    code, V, T, s = _solve(ctx, A, B, params, seed=seed, M=M, mass=M is not None, options=options)
    out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": seed}), M=M)
    assert code == out["ret"] == 0
    h, ho = s.history(), out["res_hist"]
    assert abs(h[0] - ho[0]) <= first_rtol * abs(ho[0])  # first trip: same start vector, same arithmetic up to summation order
    if trajectory:
        assert abs(s.trips() - out["trips"]) <= 1
        n = min(nhist, len(h), len(ho))
        np.testing.assert_allclose(h[:n], ho[:n], rtol=1e-6)
    else:
        assert abs(s.trips() - out["trips"]) <= max(3, 0.25 * out["trips"])
    Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) <= (10 if trajectory else 50) * tol
    return V, T, s, out


def test_config1_dense_m256_reference_parameters(ctx, oracle):
    # BASELINE configs[0]: dense stable A m=256, M=I, B m x 4, k_max=32 (the StlWrapper CPU case) with the parameters
    # of SURVEY 8(d): Lanczos iterations 10 > 2 + p = 6, so invariants only
    from rails_amd import problems as P

    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 4, seed=2)
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=1, trajectory=False)
    R = _residual(A, B, V, T)
    assert np.linalg.norm(R) / np.linalg.norm(B @ B.T) < 2e-3
    assert abs(s.relative_residual() - np.linalg.norm(R) / np.linalg.norm(B @ B.T)) < 1e-9


def _trip_by_trip_from_oracle_states(ctx, oracle, A, B, params, seed, trips, options=None, M=None):
    """Deterministic comparison where `Lanczos iterations` > 2 + p (the parameter sets of SURVEY 8(d)).  Free-running trajectories of two
    correct implementations part ways there: which Ritz directions expand the space is decided by comparisons of nearly equal numbers
    (oracle/README.md).  So the trajectory is the oracle's, and the implementation under test is asked, at every point of it, for the
    NEXT trip: from the oracle's basis after j trips (handed over through "Restart from solution", the reference's warm start) one trip
    with the same seeds -- projected solve, residual estimate, solution -- has to reproduce the oracle's own next trip."""
    worst_est, worst_x = 0.0, 0.0
    for j in range(1, trips + 1):
        head = oracle.solve(A, B, oracle.params({**params, "Maximum iterations": j, "rng_mode": 1, "seed": seed}), M=M)
        if head["ret"] == 0:
            break  # converged: the trajectory ends here
        V0 = np.ascontiguousarray(head["V"])
        k = V0.shape[1]
        assert np.abs(V0.T @ V0 - np.eye(k)).max() < 1e-12
        one = {**params, "Restart from solution": 1, "Maximum iterations": 1}
        out = oracle.solve(A, B, oracle.params({**one, "rng_mode": 1, "seed": seed + j}), M=M, V0=V0)
        code, V, T, s = _solve(ctx, A, B, one, seed=seed + j, M=M, mass=M is not None, V0=V0, options=options)
        assert code == out["ret"]
        h, ho = s.history(), out["res_hist"]
        assert len(h) == len(ho) == 1
        worst_est = max(worst_est, abs(h[0] - ho[0]) / abs(ho[0]))
        Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
        worst_x = max(worst_x, np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo))
        assert V.shape[1] == out["V"].shape[1]  # the same number of directions kept / added, restart trips included
    return j, worst_est, worst_x


@pytest.mark.parametrize("backend", ["direct", "subspace"])
def test_reference_parameter_sets_trip_by_trip(ctx, oracle, backend):
    # BASELINE configs[0] (dense m = 256, B m x 4, Restart 32 / Reduced 16 / Expand 3 / Lanczos 10: twelve trips, across the first restart)
    # and configs[1] at reduced size (7-pt Laplacian, B m x 8, Restart 64 / Reduced 32 / Expand 8 / Lanczos 20): Lanczos iterations > 2 + p
    from rails_amd import problems as P

    options = {"subspace": 1 if backend == "subspace" else 0}
    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 4, seed=2)
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3}
    j, est, x = _trip_by_trip_from_oracle_states(ctx, oracle, A, B, params, seed=1, trips=12, options=options)
    print("config 1 (%s): %d trips, residual estimates within %.1e, X within %.1e" % (backend, j, est, x))
    assert j >= 11 and est <= 1e-10 and x <= 1e-10  # measured: 3e-14 / 4e-15 (direct), 7e-14 / 2e-13 (coordinate space)
    A = P.laplace7(12, 12, 10)
    B = P.rhs(1440, 8, seed=3)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3}
    j, est, x = _trip_by_trip_from_oracle_states(ctx, oracle, A, B, params, seed=4, trips=9, options=options)
    print("config 2 (%s): %d trips, residual estimates within %.1e, X within %.1e" % (backend, j, est, x))
    # measured: 6e-14 / 6e-15 (direct), 7e-14 / 9e-15 (coordinate space).  (Round 2 had 8e-6 / 3e-9 here for the coordinate-space back end
    # and blamed the rounding of the Lanczos start vector; it was a defect of the device basis instead, isolated and fixed in round 3 --
    # see test_coordinate_space_basis_stays_orthonormal_through_degenerate_blocks below.)
    assert j >= 5 and x <= 1e-10 and est <= 1e-10


def test_coordinate_space_basis_stays_orthonormal_through_degenerate_blocks(ctx, oracle, monkeypatch):
    """What the round-2 deviation of the coordinate-space back end on configs[1] was: the first residual directions of a run lie close
    to span(B), the block of a warm-start V (or of A * V) built from them is numerically rank deficient, and the column-at-a-time
    path of the block orthogonalisation projected each column only against the columns accepted from the same block -- a column that
    keeps a fraction f of its length inherits their defects against the older basis columns magnified by 1 / f.  Along one block:
    P'P - I = 2e-14, 2e-12, 1e-10, 4e-9, 1e-8; two trips later 3e-5; the projected matrix V'AV (src/LyapunovSolver.hpp:146-160) off and
    asymmetric by the same amounts.  With RAILS_SUBSPACE_VERIFY the back end measures, block by block, |X - P c| / |X| and P'P - I on the
    device; both have to stay at rounding level on exactly that state (the oracle's V after 4 and after 6 trips)."""
    from rails_amd import problems as P

    monkeypatch.setenv("RAILS_SUBSPACE_VERIFY", "1")  # read when the solver's basis is created
    A = P.laplace7(12, 12, 10)
    B = P.rhs(1440, 8, seed=3)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3}
    one = {**params, "Restart from solution": 1, "Maximum iterations": 1}
    for j in (4, 6):
        head = oracle.solve(A, B, oracle.params({**params, "Maximum iterations": j, "rng_mode": 1, "seed": 4}))
        V0 = np.ascontiguousarray(head["V"])
        out = oracle.solve(A, B, oracle.params({**one, "rng_mode": 1, "seed": 4 + j}), V0=V0)
        code, V, T, s = _solve(ctx, A, B, one, seed=4 + j, V0=V0, options={"subspace": 1})
        st = s.backend_stats()
        print(j, {k: st[k] for k in ("verified", "verify_representation", "verify_orthonormality", "one_by_one")})
        assert st["verified"] == 1 and st["one_by_one"] >= 1  # the degenerate path is the one under test
        assert st["verify_orthonormality"] <= 1e-13  # measured 9e-16 (before the fix: 1.2e-8 and 3.4e-5)
        assert st["verify_representation"] <= 1e-11  # measured 5e-14 .. 9e-13, relative to the block's columns as handed over
        Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
        assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) <= 1e-10


def test_coordinate_space_basis_stays_orthonormal_over_a_long_run(ctx, monkeypatch):
    """The same self-check over 150 trips and two dozen restarts of a stagnating solve (7-point Laplacian, tolerance out of reach), with the
    overlapped block orthogonalisation at work: every block the device finishes behind the host's back is measured at its read-back.
    (The round-2 rule of thumb for what goes wrong here: a looser re-orthogonalisation rule gave V'V - I of 1e-14, 2e-12, 6e-7, 0.9 after
    50, 100, 200, 400 trips -- SubspaceWrappers.hpp, absorb_tail.)"""
    from rails_amd import problems as P

    monkeypatch.setenv("RAILS_SUBSPACE_VERIFY", "1")
    A = P.laplace7(30, 30, 20)
    m = A[0].size - 1
    B = P.rhs(m, 8, seed=2)
    params = {"Restart size": 96, "Reduced size": 48, "Expand size": 8, "Lanczos iterations": 12, "Tolerance": 1e-14, "Maximum iterations": 150}
    code, V, T, s = _solve(ctx, A, B, params, seed=3, options={"subspace": 1})
    st = s.backend_stats()
    print({k: st[k] for k in ("dim", "absorb", "overlapped_blocks", "one_by_one", "compress", "verify_representation", "verify_orthonormality")})
    assert code != 0 and s.trips() == 150  # (the tolerance is out of reach on purpose)
    assert st["verified"] == 1 and st["overlapped_blocks"] >= 100 and st["compress"] >= 10
    assert st["verify_orthonormality"] <= 1e-12 and st["verify_representation"] <= 1e-11  # measured 1e-14 / 1e-15
    Q = V.T @ V
    assert np.abs(Q - np.eye(Q.shape[0])).max() <= 1e-12
    s.close()


def test_config1_dense_m256_trajectory(ctx, oracle):
    # same matrix, B m x 8 and Lanczos iterations 8 <= 2 + p: the whole trajectory must match the oracle
    from rails_amd import problems as P

    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 8, seed=2)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=1, trajectory=True)
    assert np.linalg.norm(_residual(A, B, V, T)) / np.linalg.norm(B @ B.T) < 2e-3


def test_config2_laplace_small_matches_oracle(ctx, oracle):
    # BASELINE configs[1] at a size the oracle finishes in seconds: 7-pt Laplacian 20x20x15, B m x 8, k = 64
    from rails_amd import problems as P

    A = P.laplace7(20, 20, 15)
    m = A[0].size - 1
    B = P.rhs(m, 8, seed=5)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=3, trajectory=True)
    assert s.relative_residual() < 5e-3
    Q = V.T @ V
    assert np.abs(Q - np.eye(Q.shape[0])).max() < 1e-10
    # SURVEY 8(d) parameters (Lanczos iterations 20): invariants
    params["Lanczos iterations"] = 20
    _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=3, trajectory=False)


def test_generalized_mass_matrix_matches_oracle(ctx, oracle):
    # BASELINE configs[4] shape at small size: SPD diagonal mass matrix, generalized equation A X M' + M X A' + B B' = 0
    # (spec: matlab/RAILSsolver.m:368-395; parity unpinned in C++, residual-checked)
    from rails_amd import problems as P

    A = P.laplace7(12, 12, 10)
    m = A[0].size - 1
    M = P.mass_diag(m, seed=4)
    B = P.rhs(m, 6, seed=6)
    params = {"Restart size": 60, "Reduced size": 30, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-4}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-4, seed=9, M=M)
    # true generalized residual on the host
    import scipy.sparse as sp

    As = sp.csr_matrix((A[2], A[1], A[0]), shape=(m, m))
    Md = M[2]
    X = V @ T @ V.T
    R = As @ X * Md[None, :] + (Md[:, None] * X) @ As.T + B @ B.T
    assert np.linalg.norm(R) / np.linalg.norm(B @ B.T) < 2e-3
    assert abs(s.relative_residual() - np.linalg.norm(R) / np.linalg.norm(B @ B.T)) < 1e-8


def test_warm_start_continuation(ctx, oracle):
    # configs[4] "warm-start V from previous solve": perturb A's diagonal by 1%, restart from the previous V
    from rails_amd import problems as P

    A = P.laplace7(10, 10, 10)
    m = 1000
    B = P.rhs(m, 2, seed=8)
    params = {"Restart size": 60, "Reduced size": 30, "Expand size": 2, "Lanczos iterations": 10, "Tolerance": 1e-5}
    code, V, T, s = _solve(ctx, A, B, params, seed=2)
    assert code == 0
    cold_trips = s.trips()
    rowptr, col, val = A
    val2 = val.copy()
    diag = col == np.repeat(np.arange(m), np.diff(rowptr))
    val2[diag] *= 1.01
    A2 = (rowptr, col, val2)
    code, V2, T2, s2 = _solve(ctx, A2, B, {**params, "Restart from solution": 1}, seed=3, V0=V)
    assert code == 0
    assert s2.trips() < cold_trips
    assert s2.relative_residual() < 1e-3


def test_projected_lanczos_matches_oracle(ctx, oracle):
    # opt-in coefficient-space form of the residual Lanczos (rails/HipSolverOps.hpp): same random start vector, same
    # recurrence in an orthonormal basis of span[V, AV, B, q0] -> same tridiagonal matrix, same expansion vectors up to
    # rounding (the Cholesky factor of the Gram matrix of [AV, B] - V V'[AV, B] costs a few digits: 1e-7 on the estimate)
    from rails_amd import problems as P

    opts = {"projected_lanczos": 1}
    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 8, seed=2)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-3}
    before = ctx.stats()
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=1, trajectory=True, options=opts, first_rtol=1e-7)
    after = ctx.stats()
    assert after["lanczos_start"] - before["lanczos_start"] >= s.trips() - 2  # the projected form really ran
    assert np.linalg.norm(_residual(A, B, V, T)) / np.linalg.norm(B @ B.T) < 2e-3
    Q = V.T @ V
    assert np.abs(Q - np.eye(Q.shape[0])).max() < 1e-10

    A = P.laplace7(20, 20, 15)
    m = A[0].size - 1
    B = P.rhs(m, 8, seed=5)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=3, trajectory=True, options=opts, first_rtol=1e-7)
    assert s.relative_residual() < 5e-3
    # odd expand size, Lanczos past the rank of the residual operator: invariants only
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3}
    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 4, seed=2)
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=1, trajectory=False, options=opts, first_rtol=1e-7)
    assert np.linalg.norm(_residual(A, B, V, T)) / np.linalg.norm(B @ B.T) < 2e-3


def test_projected_lanczos_falls_back_with_mass_matrix(ctx, oracle):
    from rails_amd import problems as P

    A = P.laplace7(12, 12, 10)
    m = A[0].size - 1
    M = P.mass_diag(m, seed=4)
    B = P.rhs(m, 6, seed=6)
    params = {"Restart size": 60, "Reduced size": 30, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-4}
    before = ctx.stats()
    _compare_with_oracle(ctx, oracle, A, B, params, 1e-4, seed=9, M=M, options={"projected_lanczos": 1})
    assert ctx.stats()["lanczos_start"] == before["lanczos_start"]


# ---- the coordinate-space back end (rails/SubspaceWrappers.hpp, option "subspace") ------------------------------------------
SUB = {"subspace": 1}


def test_subspace_backend_known_answers(ctx):
    # the reference's KATs (test/LyapunovSolverEpetra_test.cpp:51-177) and n = 20 shapes (test/LyapunovSolver_test.cpp:118-300)
    A = np.array([[0.0, 1.0], [-5.0, -5.0]])
    for B, Xexp in ((-np.eye(2), [[0.62, -0.5], [-0.5, 0.6]]), (np.array([[-1.0], [-1.0]]), [[0.82, -0.5], [-0.5, 0.6]])):
        code, V, T, s = _solve(ctx, A, B, {"Minimize solution space": 0}, options=SUB)
        assert code == 0 and s.backend_stats()["absorb"] > 0  # the coordinate-space back end really ran
        np.testing.assert_allclose(V @ T @ V.T, Xexp, rtol=0, atol=1e-13)
    g = np.random.default_rng(1)
    n = 20
    A = g.uniform(-1, 1, (n, n))
    B = g.uniform(-1, 1, (n, 1))
    code, V, T, _ = _solve(ctx, A, B, {}, options=SUB)
    assert code == 0 and np.abs(_residual(A, B, V, T)).max() < 1e-3
    A, B = _tridiagonal_problem(20, 2)
    code, V, T, _ = _solve(ctx, A, B, {"Restart Size": 19, "Reduced Size": 15, "Expand Size": 1, "Minimize solution space": 0}, options=SUB)
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3
    A, B = _tridiagonal_problem(20, 3)
    code, V, T, _ = _solve(ctx, A, B, {"Minimize solution space": 0, "Tolerance": 1e-8}, options=SUB)
    assert code == 0 and V.shape[1] == 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3
    code, V, T, _ = _solve(ctx, A, B, {"Minimize solution space": 1, "Tolerance": 1e-8}, options=SUB)
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3
    A, B = _tridiagonal_problem(20, 4)
    code, V, T, _ = _solve(ctx, A, B, {"Restart iterations": 10, "Minimize solution space": 0, "Expand size": 1}, options=SUB)
    assert code == 0 and V.shape[1] < 20 and np.abs(_residual(A, B, V, T)).max() < 1e-3


def test_subspace_backend_matches_oracle(ctx, oracle):
    """Same seeds, same algorithm, every multivector as coordinates in one orthonormal device basis: the trajectory must match the
    oracle like the direct back end's does (first estimate 1e-9, first trips 1e-6, X to 10*tol), V must come out orthonormal."""
    from rails_amd import problems as P

    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 8, seed=2)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=1, trajectory=True, options=SUB)
    st = s.backend_stats()
    assert st["absorb"] > 0 and st["compress"] >= 1
    assert np.linalg.norm(_residual(A, B, V, T)) / np.linalg.norm(B @ B.T) < 2e-3
    assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-12
    # BASELINE configs[0] parameters (Lanczos iterations past the rank of the residual operator: invariants only)
    B4 = P.rhs(256, 4, seed=2)
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B4, params, 1e-3, seed=1, trajectory=False, options=SUB)
    assert np.linalg.norm(_residual(A, B4, V, T)) / np.linalg.norm(B4 @ B4.T) < 2e-3
    # configs[1] at oracle size
    A = P.laplace7(20, 20, 15)
    m = A[0].size - 1
    B = P.rhs(m, 8, seed=5)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-3}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-3, seed=3, trajectory=True, options=SUB)
    assert s.relative_residual() < 5e-3
    assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-12
    # tight tolerance: no noise floor (unlike the coefficient-space Lanczos of the direct back end)
    params = {"Restart size": 96, "Reduced size": 48, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-6}
    code, V, T, s = _solve(ctx, A, B, params, seed=3, options=SUB)
    assert code == 0 and s.relative_residual() < 1e-5 and abs(s.trips() - 58) <= 3  # the oracle takes 58 trips


def test_subspace_backend_mass_matrix_and_warm_start(ctx, oracle):
    """generalized equation (M * W absorbed like A * W) and warm start (the caller's V absorbed) on the coordinate-space back end"""
    from rails_amd import problems as P
    import scipy.sparse as sp

    A = P.laplace7(12, 12, 10)
    m = A[0].size - 1
    M = P.mass_diag(m, seed=4)
    B = P.rhs(m, 6, seed=6)
    params = {"Restart size": 60, "Reduced size": 30, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-4}
    V, T, s, out = _compare_with_oracle(ctx, oracle, A, B, params, 1e-4, seed=9, M=M, options=SUB)
    assert s.backend_stats()["absorb"] > 0
    As = sp.csr_matrix((A[2], A[1], A[0]), shape=(m, m))
    Md = M[2]
    X = V @ T @ V.T
    R = As @ X * Md[None, :] + (Md[:, None] * X) @ As.T + B @ B.T
    assert np.linalg.norm(R) / np.linalg.norm(B @ B.T) < 2e-3
    assert abs(s.relative_residual() - np.linalg.norm(R) / np.linalg.norm(B @ B.T)) < 1e-8
    # tridiagonal SPD mass matrix: M * W needs ghost-free but non-diagonal products
    Mt = P.mass_tridiag(m)
    code, V2, T2, s2 = _solve(ctx, A, B, params, seed=9, M=Mt, mass=True, options=SUB)
    code_d, V3, T3, s3 = _solve(ctx, A, B, params, seed=9, M=Mt, mass=True)
    assert code == code_d == 0 and s2.backend_stats()["absorb"] > 0 and s3.backend_stats() == {}
    X2, X3 = V2 @ T2 @ V2.T, V3 @ T3 @ V3.T
    assert np.linalg.norm(X2 - X3) / np.linalg.norm(X3) < 1e-3
    # warm start (test/LyapunovSolver_test.cpp:333-341 shape): perturb A's diagonal by 1 %, restart from the previous V
    A1 = P.laplace7(10, 10, 10)
    m1 = A1[0].size - 1
    B1 = P.rhs(m1, 4, seed=6)
    prm = {"Restart size": 40, "Reduced size": 20, "Expand size": 4, "Lanczos iterations": 6, "Tolerance": 1e-4}
    code, Va, Ta, sa = _solve(ctx, A1, B1, prm, seed=2, options=SUB)
    assert code == 0
    rowptr, col, val = A1
    val2 = val.copy()
    val2[col == np.repeat(np.arange(m1), np.diff(rowptr))] *= 1.01
    A2 = (rowptr, col, val2)
    code, Vb, Tb, sb = _solve(ctx, A2, B1, {**prm, "Restart from solution": 1}, seed=3, V0=Va, options=SUB)
    assert code == 0 and sb.backend_stats()["absorb"] > 0 and sb.trips() < sa.trips()
    code, Vc, Tc, sc = _solve(ctx, A2, B1, {**prm, "Restart from solution": 1}, seed=3, V0=Va)  # direct back end
    assert code == 0 and abs(sb.trips() - sc.trips()) <= 2
    Xb, Xc = Vb @ Tb @ Vb.T, Vc @ Tc @ Vc.T
    assert np.linalg.norm(Xb - Xc) / np.linalg.norm(Xc) < 1e-2
    assert sb.relative_residual() < 1e-3


def test_subspace_backend_long_stagnating_run_stays_orthonormal(ctx):
    """A tolerance the restart size cannot reach: hundreds of trips of restarts and small-survival A*W blocks.  The basis must stay
    orthonormal (a looser re-orthogonalisation rule than DGKS let V'V - I grow to 6e-7 after 200 trips and 0.9 after 400 here) and the
    stagnation level must be the direct back end's."""
    from rails_amd import problems as P

    Ad = P.dense_stable(256, seed=1)
    A = P.dense_to_csr(Ad)
    B = P.rhs(256, 4, seed=2)
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-8}
    res = {}
    for name, opts in (("subspace", SUB), ("direct", None)):
        code, V, T, s = _solve(ctx, A, B, params, seed=1, options={**(opts or {}), "max_trips": 300})
        assert code == 2 and s.trips() == 300  # bounded run, not converged
        assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-12
        res[name] = np.linalg.norm(_residual(Ad, B, V, T), 2) / np.linalg.norm(B.T @ B, 2)
        if opts:
            st = s.backend_stats()
            assert st["second_rounds"] > 100 and st["compress"] > 20
    # the level moves with the position in the restart cycle; the two back ends sit at the same place of the same cycle
    assert res["subspace"] < 5e-3 and 0.5 < res["subspace"] / res["direct"] < 2.0


def test_subspace_backend_overlap_on_and_off(oracle, monkeypatch):
    """The overlapped block orthogonalisation of the coordinate-space back end (second projection round + CholQR on the device behind the
    host's projected solve) against the same solve with every block waiting for its own orthogonalisation: the same
    trips, the same residual estimates (to 1e-8 while the trajectories have not drifted apart), the same solution; and it is really taken."""
    import rails_amd
    from rails_amd import problems as P

    A = P.laplace7(14, 12, 10)
    m = A[0].size - 1
    B = P.rhs(m, 6, seed=5)
    params = {"Restart size": 90, "Reduced size": 40, "Expand size": 6, "Lanczos iterations": 8, "Tolerance": 1e-7}
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RAILS_SUBSPACE_OVERLAP", mode)
        c = rails_amd.Context(device=0, seed=1)
        code, V, T, s = _solve(c, A, B, params, seed=9, options={"subspace": 1})
        assert code == 0
        runs[mode] = (s.trips(), np.array(s.history()), V @ T @ V.T, s.backend_stats())
        s.close()
        c.close()
    (t0, h0, X0, st0), (t1, h1, X1, st1) = runs["0"], runs["1"]
    assert st0["overlapped_blocks"] == 0 and st1["overlapped_blocks"] >= t1 - 2
    assert t0 == t1
    # the first trips to rounding level (the prediction is good to ~1e-12); later ones as two free-running trajectories agree
    np.testing.assert_allclose(h1[:8], h0[:8], rtol=1e-8)
    np.testing.assert_allclose(h1, h0, rtol=5e-2)
    assert np.linalg.norm(X1 - X0) <= 10 * params["Tolerance"] * np.linalg.norm(X0)
    out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 9}))
    Xo = out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(X1 - Xo) <= 10 * params["Tolerance"] * np.linalg.norm(Xo)


def test_subspace_backend_rank_deficient_expansion_blocks(ctx):
    """A = -I + 0.3 x y': every A*W block is -W (already in the basis) plus multiples of ONE new direction -- nearly dependent columns whose
    projections are parallel.  Such blocks fail the conditions of the overlapped orthogonalisation (predicted Cholesky factor) and take the
    synchronous careful path; the basis stays orthonormal, nothing is absorbed twice, and the solution satisfies the equation."""
    from rails_amd import problems as P

    n = 256
    g = np.random.default_rng(12)
    x, y = g.standard_normal(n), g.standard_normal(n)
    x /= np.linalg.norm(x)
    y /= np.linalg.norm(y)
    Ad = -np.eye(n) + 0.3 * np.outer(x, y)
    B = P.rhs(n, 4, seed=3)
    params = {"Restart size": 40, "Reduced size": 20, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-9}
    code, V, T, s = _solve(ctx, P.dense_to_csr(Ad), B, params, seed=4, options={"subspace": 1})
    assert code == 0
    st = s.backend_stats()
    # span[B, x] has 4 + 1 directions (+ the random start vectors of the Lanczos runs): most of every A*W block is dropped, and the
    # expansion vectors that lie in span(V) already are replaced by random directions (the reference normalises rounding noise there)
    assert st["dropped"] > 0 and st["replaced_columns"] > 0 and st["dim"] < 40
    assert s.trips() <= 6  # the direct back end and the oracle: 4
    assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-12
    R = _residual(Ad, B, V, T)
    assert np.linalg.norm(R, 2) <= 1e-8 * np.linalg.norm(B.T @ B, 2)


def test_subspace_backend_rejected_prediction_is_loud_and_ends_the_run():
    """The read-back of an overlapped block checks that the device met the block the host booked.  With the test hook spoiling one
    prediction by 1 % the solve must stop at once and report a device failure (RAILS_EHIP through the C ABI), not run on to `Maximum
    iterations` on coordinates that no longer describe the device's vectors."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np, rails_amd\n"
        "from rails_amd import problems as P\n"
        "ctx = rails_amd.Context(device=0, seed=1)\n"
        "A = P.laplace7(14, 12, 10)\n"
        "op = rails_amd.HipOperatorWrapper(ctx, *A)\n"
        "s = rails_amd.Solver(ctx, op, P.rhs(14 * 12 * 10, 6, seed=5))\n"
        "s.set_parameters({'Restart size': 90, 'Reduced size': 40, 'Expand size': 6, 'Lanczos iterations': 8, 'Tolerance': 1e-12, 'Maximum iterations': 500})\n"
        "s.set_option('verbose', 0)\n"
        "s.set_option('subspace', 1)\n"
        "try:\n"
        "    s.solve()\n"
        "    print('RAN ON', s.trips())\n"
        "except rails_amd.RailsError as e:\n"
        "    print('FAILED LOUDLY after', s.trips(), 'trips:', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RAILS_SUBSPACE_TEST_SPOIL_PREDICTION="3", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert "FAILED LOUDLY after" in out.stdout, out.stdout + out.stderr
    trips = int(out.stdout.split("FAILED LOUDLY after")[1].split()[0])
    assert trips <= 6, out.stdout  # the third overlapped block is read back at the fourth or fifth trip: the run ends there
    assert "did not confirm its prediction" in out.stderr
