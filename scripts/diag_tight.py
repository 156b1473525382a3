"""diagnostic (not a test): tight tolerances on both back ends -- trips, residuals, distance of the two solutions"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from rails_amd import problems as P  # noqa: E402

ctx = rails_amd.Context(device=0, seed=1)


def run(name, A, B, prm, M=None):
    out = {}
    for sub in (1, 0):
        ctx.set_seed(1, 0)
        op = rails_amd.HipOperatorWrapper(ctx, *A)
        mop = rails_amd.HipOperatorWrapper(ctx, *M) if M is not None else None
        s = rails_amd.Solver(ctx, op, B, M=mop)
        assert s.set_parameters(prm) == 0
        s.set_option("verbose", 0)
        s.set_option("max_trips", 600)
        s.set_option("subspace", sub)
        if M is not None:
            s.set_option("mass", 1)
        code, V, T = s.solve()
        out[sub] = (code, V, T, s.trips(), s.relative_residual(), s.backend_stats())
        s.close()
    g = np.random.default_rng(1)
    Z = g.standard_normal((B.shape[0], 3))
    X1 = out[1][1] @ (out[1][2] @ (out[1][1].T @ Z))
    X0 = out[0][1] @ (out[0][2] @ (out[0][1].T @ Z))
    print(json.dumps({"case": name, "subspace": {"code": out[1][0], "trips": out[1][3], "k": out[1][1].shape[1], "res": out[1][4]},
                      "direct": {"code": out[0][0], "trips": out[0][3], "k": out[0][1].shape[1], "res": out[0][4]},
                      "rel_diff_X_probes": float(np.linalg.norm(X1 - X0) / np.linalg.norm(X0)),
                      "stats": {k: v for k, v in out[1][5].items() if k != "seconds"}}), flush=True)


A = P.laplace7(50, 50, 40)
run("C2 laplace 100k, restart 300/150, tol 1e-10", A, P.rhs(100000, 8, seed=3), {"Restart size": 300, "Reduced size": 150, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-10})
A = P.laplace7(30, 30, 30)
run("laplace 27k + tridiagonal mass, tol 1e-9", A, P.rhs(27000, 4, seed=5), {"Restart size": 200, "Reduced size": 100, "Expand size": 4, "Lanczos iterations": 12, "Tolerance": 1e-9}, M=P.mass_tridiag(27000))
A = P.banded_random(200000, 27, 4096, seed=2)
run("banded 200k, restart 240/120, tol 1e-11", A, P.rhs(200000, 6, seed=6), {"Restart size": 240, "Reduced size": 120, "Expand size": 6, "Lanczos iterations": 16, "Tolerance": 1e-11})
A = P.uniform_random(100000, 11, seed=2)
run("uniform-random 100k, no restart size, tol 1e-9", A, P.rhs(100000, 3, seed=7), {"Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-9})
