"""BASELINE configurations that had no `-m gpu` solve before: configs[3]'s per-rank share as a SOLVE (200 x 200 x 25 slab of the
27-point stencil, B m x 32, Restart 256 / Reduced 128 / Expand 32 / Lanczos 40), configs[1] at its full size (7-point
Laplacian 50 x 50 x 40, B m x 8, k = 64), a direct-back-end restart that keeps more than 256 vectors, and a failing device
operation inside a solve.  The solves are checked through size-independent properties, each back end against the other:
V'V = I, T = T', the reference's convergence criterion (src/LyapunovSolver.hpp:223) re-evaluated independently by power
iteration on R = A V T V' + V T V' A' + B B', agreement of the two back ends on random probes of X = V T V'.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=5)
    yield c
    c.close()


def _fro2(X):
    return float(np.trace(X.dot(X)))


def _residual_norm(ctx, op, B, V, T, steps=30):
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    Vd, Bd = MV(ctx, data=V), MV(ctx, data=B)
    AV = op.apply(Vd)
    z = MV(ctx, m=V.shape[0], n=1)
    z.random()
    lam = 0.0
    for _ in range(steps):
        z *= 1.0 / np.sqrt(_fro2(z))
        y = AV.matmul(T @ Vd.dot(z))
        y += Vd.matmul(T @ AV.dot(z))
        y += Bd.matmul(Bd.dot(z))
        lam = np.sqrt(_fro2(y))
        z = y
    return lam


def _solve_both(ctx, op, B, params, probes=4, tol_factor=2.0):
    import rails_amd
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    m = B.shape[0]
    r0 = float(np.linalg.norm(B.T @ B, 2))
    out = {}
    for subspace in (1, 0):
        ctx.set_seed(1, 0)
        s = rails_amd.Solver(ctx, op, B)
        assert s.set_parameters(params) == 0
        s.set_option("verbose", 0)
        s.set_option("subspace", subspace)
        code, V, T = s.solve()
        assert code == 0
        k = V.shape[1]
        assert k <= params["Restart size"]
        assert np.abs(T - T.T).max() <= 1e-12 * np.abs(T).max()
        Vd = MV(ctx, data=V)
        assert np.abs(Vd.dot(Vd) - np.eye(k)).max() <= 1e-10
        # Frobenius-norm residual relative to ||B B'||_F: a different measure than the solver's 2-norm criterion (checked next), same size
        assert s.relative_residual() < 2.0 * params["Tolerance"]
        assert _residual_norm(ctx, op, B, V, T) < tol_factor * params["Tolerance"] * r0
        out[subspace] = (V, T, s.trips())
        s.close()
    Z = np.random.default_rng(9).standard_normal((m, probes))
    Xz = {b: V @ (T @ (V.T @ Z)) for b, (V, T, _) in out.items()}
    assert np.linalg.norm(Xz[1] - Xz[0]) / np.linalg.norm(Xz[0]) <= 20 * params["Tolerance"]
    return out


def test_config3_per_rank_slab_solves_on_both_back_ends(ctx):
    """configs[3] (8 GPUs, m = 8M): one rank's 200 x 200 x 25 slab as a problem of its own, with that configuration's parameters"""
    import rails_amd
    from rails_amd import problems as P

    A = P.stencil27(200, 200, 25)
    m = A[0].size - 1
    assert m == 1000000
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    B = P.rhs(m, 32, seed=7)
    params = {"Restart size": 256, "Reduced size": 128, "Expand size": 32, "Lanczos iterations": 40, "Tolerance": 1e-3}
    out = _solve_both(ctx, op, B, params)
    assert abs(out[1][2] - out[0][2]) <= 3


def test_config1_full_size_solves_on_both_back_ends(ctx, oracle):
    """configs[1]: 7-point Laplacian 50 x 50 x 40 (m = 100k), B m x 8, k = 64 -- small enough for the oracle as well"""
    import rails_amd
    from rails_amd import problems as P

    A = P.laplace7(50, 50, 40)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    B = P.rhs(m, 8, seed=7)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3}
    out = _solve_both(ctx, op, B, params)
    ref = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 1}), vcap=80)
    assert ref["ret"] == 0
    Z = np.random.default_rng(2).standard_normal((m, 4))
    Xo = ref["V"] @ (ref["T"] @ (ref["V"].T @ Z))
    for b, (V, T, trips) in out.items():
        Xg = V @ (T @ (V.T @ Z))
        assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) <= 20 * params["Tolerance"]
        assert abs(trips - ref["trips"]) <= max(3, ref["trips"] // 4)


def test_direct_back_end_restart_that_keeps_more_than_256_vectors(ctx, oracle):
    """V <- V X with more than 256 columns of X (one rails_panel_gemm call cannot do that: the sliced form has to be taken;
    before, the restart silently zeroed V, AV and the run went on)"""
    import rails_amd
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A = P.laplace7(24, 20, 16)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    B = P.rhs(m, 12, seed=3)
    params = {"Restart size": 336, "Reduced size": 300, "Expand size": 12, "Lanczos iterations": 14, "Tolerance": 1e-9, "Restart tolerance": 1e-300,
              "Maximum iterations": 40}
    ctx.set_seed(4, 0)
    s = rails_amd.Solver(ctx, op, B)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    s.set_option("subspace", 0)
    code, V, T = s.solve()
    trips = s.trips()
    hist = s.history()
    s.close()
    assert code in (0, -1) and trips >= 30  # ran through the restart at 336 columns (28 trips of 12)
    k = V.shape[1]
    assert 300 <= k <= 336
    Vd = MV(ctx, data=V)
    assert np.abs(Vd.dot(Vd) - np.eye(k)).max() <= 1e-9  # zeroed panels would not be orthonormal
    ref = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 4}), vcap=348)
    assert ref["trips"] == trips
    # residual estimates right after the restart: the restarted space carries on where the oracle's does (the two trajectories have
    # drifted apart by rounding over 28 trips -- oracle/README.md -- hence percent level, not digits)
    assert np.allclose(hist[28:32], ref["res_hist"][28:32], rtol=5e-2)


def test_a_failing_device_operation_inside_a_solve_is_an_error_return(ctx):
    import rails_amd
    from rails_amd import problems as P

    A = P.laplace7(10, 8, 6)
    m = A[0].size - 1
    inner = rails_amd.HipOperatorWrapper(ctx, *A)
    calls = {"n": 0}

    def apply(trans, X, Y):
        calls["n"] += 1
        if calls["n"] == 3:
            return 1  # the third product fails
        (inner.transpose() if trans else inner).apply(X, Y)
        return 0

    op = rails_amd.HipOperatorWrapper.from_callback(ctx, m, apply)
    B = P.rhs(m, 3, seed=1)
    for subspace in (0, 1):
        calls["n"] = 0
        s = rails_amd.Solver(ctx, op, B)
        assert s.set_parameters({"Restart size": 40, "Reduced size": 20, "Expand size": 3, "Lanczos iterations": 8, "Maximum iterations": 12}) == 0
        s.set_option("verbose", 0)
        s.set_option("subspace", subspace)
        with pytest.raises(rails_amd.RailsError):
            s.solve()
        s.close()
