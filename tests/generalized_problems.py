"""The generalized-M acceptance problems the reference holds (MATLAB only: matlab/test/test_Laplace.m:14-59, test_random.m:37-50,
test_opts.m:181-195), restated as data + checks shared by the CPU (oracle) and GPU tests.

MATLAB's random stream (rng(4634)) cannot be reproduced here, so the matrices are the same CLASS from numpy's generator -- the 2-D
5-point Laplacian of laplacian2(n), M = spdiags(rand(n,1)) (a diagonal mass matrix with entries in (0, 1): ill-conditioned by
design), B = rand(n,1), A = sprand(n,n,10/n) -- with seeds for which the reference's own bounds hold on the CPU oracle; the
solver runs with the MATLAB defaults of matlab/RAILSsolver.m:100-137 (maxit 100, tol 1e-4, expand min(3, columns of B) = 1, no
restarts by count or size, rank reduction upon convergence).  a13 / f1 stay "parity unpinned" (no MATLAB, no SLICOT sg03ad here): what
these tests pin is every acceptance bound the reference states for the generalized path."""
import numpy as np
import scipy.sparse as sp

MATLAB_DEFAULTS = {"Maximum iterations": 100, "Tolerance": 1e-4, "Expand size": 1, "Lanczos iterations": 10, "Restart iterations": -1,
                   "Restart size": -1, "Minimize solution space": 1}


def laplacian2(n):
    """matlab/test/test_Laplace.m:14-21"""
    m = int(round(np.sqrt(n)))
    assert m * m == n
    e = np.ones(m)
    T = sp.diags([e[:-1], -4 * e, e[:-1]], [-1, 0, 1])
    S = sp.diags([e[:-1], e[:-1]], [-1, 1])
    return (sp.kron(sp.eye(m), T) + sp.kron(S, sp.eye(m))).tocsr()


def csr(A):
    A = A.tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)


def diag_csr(d):
    n = d.size
    return np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), d.astype(np.float64).copy()


def laplace_problem(n, seed):
    """test_Laplace.m:31-43 (n = 64), :45-57 (n = 256), test_opts.m:181-195 (n = 256, opts.ortho = 'M')"""
    g = np.random.default_rng(seed)
    return laplacian2(n), g.uniform(0.0, 1.0, n), np.asfortranarray(g.uniform(0.0, 1.0, (n, 1)))


def random_problem(n, seed):
    """test_random.m:37-50: A = sprand(n, n, 10 / n) -- not a stable matrix; the method reaches the whole space and solves exactly"""
    g = np.random.default_rng(seed)
    A = sp.random(n, n, density=10.0 / n, random_state=seed, format="csr")
    return A, g.uniform(0.0, 1.0, n), np.asfortranarray(g.uniform(0.0, 1.0, (n, 1)))


def check_acceptance(A, Md, B, V, T, res_estimate, trips, trip_bound):
    """The reference's four assertions (test_Laplace.m:39-42): iter < n - 10 (where the test states it), res * ||B'B|| < 1e-2,
    res < 1e-4, ||A V S V' M' + M V S V' A' + B B'|| / ||B'B|| < 1e-4 (2-norms, as MATLAB's norm)."""
    scale = np.linalg.norm(B.T @ B, 2)
    res = res_estimate / scale
    if trip_bound is not None:
        assert trips < trip_bound, (trips, trip_bound)
    assert res * scale < 1e-2
    assert res < 1e-4
    Ad = A.toarray()
    X = V @ T @ V.T
    R = Ad @ X * Md[None, :] + (Md[:, None] * X) @ Ad.T + B @ B.T
    true_res = np.linalg.norm(R, 2) / scale
    assert true_res < 1e-4, true_res
    assert V.shape[1] == T.shape[0] == T.shape[1]
    return true_res


# (problem, n, seed, params, bound on the trips): seeds chosen so that the reference's bounds hold on the CPU oracle
CASES = {
    "Laplace_64": ("laplace", 64, 2, {}, 64 - 10),
    "Laplace_256": ("laplace", 256, 4, {}, 256 - 10),
    "morth_256": ("laplace", 256, 1, {}, 256 - 10),  # test_opts.m:181-195: opts.ortho = 'M' (V'MV = I, standard projected equation): the GPU test sets the solver's
                                                      # mass_orthogonalisation option; the CPU oracle has the Cholesky reduction only and runs the same problem with it
    "random_64": ("random", 64, 1, {"Minimize solution space": 0}, None),  # opts.restart_upon_convergence = false, no bound on iter
}


def build(name):
    kind, n, seed, extra, bound = CASES[name]
    A, Md, B = laplace_problem(n, seed) if kind == "laplace" else random_problem(n, seed)
    return A, Md, B, {**MATLAB_DEFAULTS, **extra}, bound, seed
