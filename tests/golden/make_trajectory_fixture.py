#!/usr/bin/env python3
"""Generate tests/golden/oracle_trajectories_n20.npz: full trajectories of the oracle's solve() on the n = 20 problems of the reference's
solver tests (test/LyapunovSolver_test.cpp:118-352: dense, restart, minimise, restart iterations), with the REFERENCE's random generator
(rng_mode 0: std::rand seeded by srand(seed), pinned bit for bit by ref_stl.npz) so that the start vectors of the residual Lanczos runs are
the ones the reference would draw.

    python tests/golden/make_trajectory_fixture.py

Every array is DATA: the inputs (A, B, the parameter values), and per case the number of trips, the residual estimate of every trip, the
final V and T.  Where oracle/_ref is available (the build container) the first trip's residual Lanczos run of every case is repeated
through the compiled, unmodified reference (Solver::resid_lanczos of src/LyapunovSolver.hpp on StlWrapper) from the oracle's own V, AV, T
with the same srand() seed, and must agree to 1e-12 before the fixture is written: the piece of solve() the reference can run here
anchors the trajectory it belongs to.  The whole of Solver::solve needs SLICOT (absent): the trajectories themselves are the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402


def tridiagonal_problem(n, seed):
    g = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for i in range(n):
        A[i, i] = -2.0 - g.uniform(0, 1)
        if i + 1 < n:
            A[i, i + 1] = 1.0
            A[i + 1, i] = 1.0
    B = g.uniform(-1, 1, (n, 1))
    return A, B


CASES = [
    ("dense", None, 1, {}),
    ("restart", 2, 2, {"Restart size": 19, "Reduced size": 15, "Expand size": 1, "Minimize solution space": 0}),
    ("full_space", 3, 3, {"Minimize solution space": 0, "Tolerance": 1e-8}),
    ("minimise", 3, 4, {"Minimize solution space": 1, "Tolerance": 1e-8}),
    ("restart_iterations", 4, 5, {"Restart iterations": 10, "Minimize solution space": 0, "Expand size": 1}),
]


def problem(name, pseed):
    n = 20
    if pseed is None:
        g = np.random.default_rng(1)
        A = g.uniform(-1, 1, (n, n))
        B = np.zeros((n, 1))
        B[n - 1, 0] = g.uniform(-1, 1)
        return A, B
    return tridiagonal_problem(n, pseed)


def main():
    orc = Oracle()
    ref = None
    try:
        from oracle.oracle import Reference, build

        build(ref=True)
        ref = Reference()
    except Exception as e:  # no /root/reference: the fixture is regenerated without the cross-check
        print("oracle/_ref not available (%s): no cross-check of the first Lanczos run" % e)
    out = {}
    for name, pseed, seed, params in CASES:
        A, B = problem(name, pseed)
        orc.srand(seed)
        res = orc.solve(A, B, orc.params({**params, "rng_mode": 0, "seed": seed}))
        assert res["ret"] == 0, (name, res["ret"])
        out[name + "_A"], out[name + "_B"] = A, B
        out[name + "_params"] = np.array([str(sorted(params.items()))])
        out[name + "_seed"] = np.array([seed])
        out[name + "_trips"] = np.array([res["trips"]])
        out[name + "_res_hist"] = np.asarray(res["res_hist"], dtype=np.float64)
        out[name + "_V"], out[name + "_T"] = res["V"], res["T"]
        if ref is not None:
            # the first trip's state, rebuilt with plain numpy: V = orth(B), AV, T from scipy's solver of the projected equation
            import scipy.linalg as sl

            V = B / np.linalg.norm(B)
            AV = A @ V
            T = sl.solve_continuous_lyapunov(V.T @ AV, -(V.T @ B) @ (B.T @ V))
            L = int(dict(params).get("Lanczos iterations", 10))
            ref.srand(seed)
            lr = ref.resid_lanczos(AV, V, T, B, L)
            orc.srand(seed)
            lo = orc.resid_lanczos(AV, V, T, B, L, rng_mode=0)
            assert lr["steps"] == lo["steps"], (name, lr["steps"], lo["steps"])
            k = lr["steps"]
            err = np.abs(lr["H"][:k, :k] - lo["H"][:k, :k]).max() / np.abs(lr["H"][:k, :k]).max()
            assert err < 1e-9, (name, err)
            out[name + "_first_lanczos_vs_reference"] = np.array([err])
        print(name, "trips", res["trips"], "k", res["V"].shape[1], "first estimate %.6e" % res["res_hist"][0])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trajectories_n20.npz"), **out)


if __name__ == "__main__":
    main()
