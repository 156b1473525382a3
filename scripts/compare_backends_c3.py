#!/usr/bin/env python3
"""C3 (m = 1M banded-random, B m x 16, Restart 200 / Reduced 128 / Expand 16 / Lanczos 20) solved to a tolerance on both back ends:
trip counts, residuals, and the distance between the two solutions X = V T V' measured through their action on random vectors
(X is 1M x 1M and never formed).  usage: python scripts/compare_backends_c3.py [tolerance]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from rails_amd import problems as P  # noqa: E402

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
m = 1000000
A = P.banded_random(m, 27, 4096, seed=1)
B = P.rhs(m, 16, seed=8)
ctx = rails_amd.Context(device=0, seed=1)
op = rails_amd.HipOperatorWrapper(ctx, *A)
res = {}
for name, sub in (("direct", 0), ("coordinates", 1)):
    ctx.set_seed(1, 0)
    s = rails_amd.Solver(ctx, op, B)
    assert s.set_parameters({"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": tol}) == 0
    s.set_option("verbose", 0)
    s.set_option("subspace", sub)
    t0 = time.perf_counter()
    code, V, T = s.solve()
    dt = time.perf_counter() - t0
    res[name] = dict(code=code, trips=s.trips(), k=V.shape[1], seconds=dt, rel=s.relative_residual(), V=V, T=T, hist=s.history())
    s.close()
g = np.random.default_rng(5)
Z = g.standard_normal((m, 4))
Y = {n: r["V"] @ (r["T"] @ (r["V"].T @ Z)) for n, r in res.items()}
d = np.linalg.norm(Y["direct"] - Y["coordinates"]) / np.linalg.norm(Y["direct"])
out = {"tolerance": tol, "|X_direct z - X_coordinates z| / |X_direct z| (4 random z)": d}
for n, r in res.items():
    out[n] = {k: r[k] for k in ("code", "trips", "k", "seconds", "rel")}
    out[n]["first_estimates"] = [float(x) for x in r["hist"][:6]]
    out[n]["orthonormality |V'V - I|_max"] = float(np.abs(r["V"].T @ r["V"] - np.eye(r["k"])).max())
print(json.dumps(out, indent=1))
ctx.close()
