"""The reference's application problem end to end on the GPU: the MOC ocean model of matlab/DataErik/ (tests/golden/moc_erik.npz) as
matlab/test/test_MOC.m:12-36 runs it -- border, Schur complement on the unknowns with a nonzero mass entry, generalized RAILS solve
at tolerance 1e-3 -- with the reference's two acceptance checks: the residual of the reduced equation and, after mapping the solution
back to all unknowns (RAILSschur.m's Vtrans), the residual of the original descriptor equation, both < 1e-3 in the Frobenius norm."""
import numpy as np
import pytest

from moc_problem import add_border, load, schur_dense

pytestmark = pytest.mark.gpu

PARAMS = {"Maximum iterations": 1000, "Tolerance": 1e-3, "Expand size": 3, "Lanczos iterations": 10}


@pytest.mark.parametrize("subspace,device_solve", [(1, False), (0, False), (1, True), (0, True)])
def test_moc_schur_generalized_solve(subspace, device_solve, oracle):
    import rails_amd
    from rails_amd import problems as P
    from rails_amd.schur import SchurOperator

    A, mdiag, B = load()
    n = A.shape[0]
    A2, m2, B2 = add_border(A, mdiag, B)
    ctx = rails_amd.Context(device=0, seed=1)
    S = SchurOperator(ctx, (A2.indptr.astype(np.int64), A2.indices.astype(np.int32), A2.data.astype(np.float64)), m2, tol=1e-12, device_solve=device_solve)
    Sd, ms, BSd, i1, i2 = schur_dense(A2, m2, B2)
    assert S.m2 == 512 and np.array_equal(S.idx2, i2)
    np.testing.assert_allclose(S.dense(), Sd, atol=1e-10 * np.abs(Sd).max())
    BS = S.restrict(B2)
    np.testing.assert_array_equal(BS, BSd)
    assert np.all(S.mass22 < 0)  # this data set's mass entries are negative: the solver flips (S, M) -> (-S, -M) in the projected solve
    Mop = rails_amd.HipOperatorWrapper(ctx, np.arange(S.m2 + 1, dtype=np.int64), np.arange(S.m2, dtype=np.int32), S.mass22)
    s = rails_amd.Solver(ctx, S.op, BS, M=Mop)
    assert s.set_parameters(PARAMS) == 0
    s.set_option("verbose", 0)
    s.set_option("mass", 1)
    s.set_option("subspace", subspace)
    code, V, T = s.solve()
    assert code == 0
    X = V @ T @ V.T
    # test_MOC.m:30-31
    R = Sd @ X * ms[None, :] + (ms[:, None] * X) @ Sd.T + BSd @ BSd.T
    assert np.linalg.norm(R) < 1e-3
    # test_MOC.m:33-36: back to all unknowns, the original equation
    Vf = S.prolongate(V)[:n]
    Ad = A.toarray()
    Xf = Vf @ T @ Vf.T
    Rf = Ad @ Xf * mdiag[None, :] + (mdiag[:, None] * Xf) @ Ad.T + B @ B.T
    assert np.linalg.norm(Rf) < 1e-3
    # and the CPU oracle on the same reduced problem from the same seeds
    m = S.m2
    out = oracle.solve(P.dense_to_csr(Sd), BSd, oracle.params({**PARAMS, "rng_mode": 1, "seed": 1}),
                       M=(np.arange(m + 1, dtype=np.int64), np.arange(m, dtype=np.int32), ms.copy()))
    assert out["ret"] == 0
    Xo = out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 50 * PARAMS["Tolerance"]
    # Several hundred trips of a slowly converging solve with `Lanczos iterations` (10) > 2 + p: the chaotic regime of the reference's own
    # recurrence (oracle/README.md), in which the residual estimates of two implementations part from the first trips on (measured: they
    # differ by more than 1e-6 at trip 1 on both back ends), so the trajectories cannot be compared trip by trip; what they reach agrees
    # (above).  Their lengths, measured (tests/diag_bounds.py): oracle 575 trips, direct back end 499 (0.87x), coordinate-space back
    # end 722 (1.26x) with the A11 solve on the host -- asserted with a margin of 10 % on those ratios; with the solve on the device (the
    # same factors applied in another order of operations: a perturbation at rounding level) 541 (0.94x) and 862 (1.50x): bound 1.6.
    print("MOC trips: oracle %d, subspace %d device_solve %d: %d; A11 levels %s" % (out["trips"], subspace, device_solve, s.trips(), S.dlu.levels() if S.dlu else None))
    bound = 1.6 if device_solve else 1.4
    assert out["trips"] / bound <= s.trips() <= bound * out["trips"]
    s.close()
    ctx.close()
