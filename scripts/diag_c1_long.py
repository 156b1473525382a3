"""diagnostic (not a test): C1 at a tolerance the restart size cannot reach -- long stagnating runs on both back ends"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from rails_amd import problems as P  # noqa: E402

ctx = rails_amd.Context(device=0, seed=1)
Ad = P.dense_stable(256, seed=1)
A = P.dense_to_csr(Ad)
B = P.rhs(256, 4, seed=2)
prm = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": float(os.environ.get("TOL", "1e-8"))}
for sub in (1, 0):
    for mt in (25, 50, 100, 200, 400):
        ctx.set_seed(1, 0)
        op = rails_amd.HipOperatorWrapper(ctx, *A)
        s = rails_amd.Solver(ctx, op, B)
        assert s.set_parameters(prm) == 0
        s.set_option("verbose", 0)
        s.set_option("max_trips", mt)
        s.set_option("subspace", sub)
        code, V, T = s.solve()
        X = V @ T @ V.T
        R = Ad @ X + X @ Ad.T + B @ B.T
        h = s.history()
        print(json.dumps({"subspace": sub, "max_trips": mt, "code": code, "trips": s.trips(), "k": V.shape[1], "true_rel_res": float(np.linalg.norm(R, 2) / np.linalg.norm(B.T @ B, 2)),
                          "VtV-I": float(np.abs(V.T @ V - np.eye(V.shape[1])).max()), "T_sym": float(np.abs(T - T.T).max()), "last_estimates": [float(x) for x in h[-3:]],
                          "backend": {k: v for k, v in s.backend_stats().items() if k != "seconds"}}), flush=True)
        s.close()
