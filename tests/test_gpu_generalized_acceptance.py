"""The reference's generalized-M acceptance tests (matlab/test/test_Laplace.m:31-59, test_random.m:37-50, test_opts.m:181-195; restated
in tests/generalized_problems.py) on the HIP path, both back ends, through the C ABI: the same four bounds the reference asserts, and
agreement with the CPU oracle from the same seeds."""
import numpy as np
import pytest

import generalized_problems as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("subspace", [1, 0])
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_generalized_acceptance(oracle, name, subspace):
    import rails_amd

    A, Md, B, params, bound, seed = G.build(name)
    ctx = rails_amd.Context(device=0, seed=seed)
    op = rails_amd.HipOperatorWrapper(ctx, *G.csr(A))
    mop = rails_amd.HipOperatorWrapper(ctx, *G.diag_csr(Md))
    s = rails_amd.Solver(ctx, op, B, M=mop)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    s.set_option("mass", 1)
    s.set_option("subspace", subspace)
    morth = name.startswith("morth")
    if morth:
        s.set_option("mass_orthogonalisation", 1)  # opts.ortho = 'M' (matlab/RAILSsolver.m:583-597): V'MV = I, `lyap(VAV, VBV)`
    code, V, T = s.solve()
    assert code == 0
    if morth:
        assert np.abs(V.T @ (Md[:, None] * V) - np.eye(V.shape[1])).max() < 1e-10
    true_res = G.check_acceptance(A, Md, B, V, T, abs(s.history()[-1]), s.trips(), bound)
    # the library's own evaluation of the generalized residual (Frobenius norms, from panel products without forming X: the squares of
    # three terms combined, so its floor is ~sqrt(eps) of ||B B'||) agrees with the dense one
    Ad = A.toarray()
    X = V @ T @ V.T
    R = Ad @ X * Md[None, :] + (Md[:, None] * X) @ Ad.T + B @ B.T
    assert abs(s.relative_residual() - np.linalg.norm(R) / np.linalg.norm(B @ B.T)) < 1e-6
    # the CPU oracle from the same seeds: the same solution to the tolerance of the solve, a trajectory of comparable length
    out = oracle.solve(G.csr(A), B, oracle.params({**params, "rng_mode": 1, "seed": seed}), M=G.diag_csr(Md))
    assert out["ret"] == 0
    Xo = out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(X - Xo) <= 50 * params["Tolerance"] * np.linalg.norm(Xo)
    if not morth:  # (the oracle runs the other formulation of that case: another trajectory to the same solution)
        assert abs(s.trips() - int(out["trips"])) <= max(3, int(out["trips"]) // 5)
    s.close()
    ctx.close()
